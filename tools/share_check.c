/* share_check.c -- a --devices worker's way into a BAM: the header through one reader (closed at once, its read-ahead still in
 * flight), the records from the virtual offset the .bai gives for (contig 0, cut).  Built with sanitizers by tools/sanitize_host.sh.
 * Usage: share_check reads.bam cut */
#include <stdio.h>
#include <stdlib.h>
#include "mmhost.h"
#include "bamio.h"
int main(int argc,char**argv){
  const char*bam=argv[1]; char bai[4096]; snprintf(bai,sizeof bai,"%s.bai",bam);
  mm_bai_t*ix=mm_bai_load(bai); if(!ix){fprintf(stderr,"no bai\n");return 1;}
  int64_t cut=atoll(argv[2]); uint64_t vo=mm_bai_start(ix,0,cut); mm_bai_free(ix);
  fprintf(stderr,"voffset %llx\n",(unsigned long long)vo);
  mmh_loader_t*ld=mmh_loader_open_share(bam,4,512,20000000,0,0,vo,0,cut,0,4<<20,0,1);
  if(!ld){fprintf(stderr,"open failed\n");return 1;}
  int more=1,set=0;long n=0;mm_batch_t b;
  while(more){int k=mmh_loader_next(ld,set,&b,&more);if(k<0){fprintf(stderr,"err\n");return 1;}n+=k;set^=1;}
  printf("reads %ld\n",n); mmh_loader_close(ld); return 0;}
