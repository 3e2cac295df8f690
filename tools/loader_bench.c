/* loader_bench.c -- time (and, built with sanitizers, check) the C ingestion path alone: BGZF inflate pool, record framing,
 * parallel flattening.  Usage: loader_bench reads.bam threads.  See tools/sanitize_host.sh. */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "mmhost.h"
static double now(void){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+t.tv_nsec*1e-9;}
int main(int argc,char**argv){
  int th=atoi(argv[2]);
  for(int rep=0;rep<2;rep++){
  double t0=now();
  mmh_loader_t*ld=mmh_loader_open(argv[1],th,4096,1000000000,0,0);
  int more=1,set=0;long n=0;mm_batch_t b;
  while(more){int k=mmh_loader_next(ld,set,&b,&more);n+=k;set^=1;}
  double t1=now();
  printf("threads %d reads %ld sec %.3f bases %.1fM -> %.1f Mbases/s\n",th,n,t1-t0,ld->processed_bases/1e6,ld->processed_bases/(t1-t0)/1e6);
  mmh_loader_close(ld);}
  return 0;}
