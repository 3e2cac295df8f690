#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "mmhost.h"
static double now(void){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+1e-9*t.tv_nsec;}
static double cpu(void){struct timespec t;clock_gettime(CLOCK_THREAD_CPUTIME_ID,&t);return t.tv_sec+1e-9*t.tv_nsec;}
int main(int argc,char**argv){
  int threads=atoi(argv[2]);
  double t0=now(),c0=cpu();
  mmh_loader_t*ld=mmh_loader_open(argv[1],threads,4096,200000000,0,0);
  int more=1,set=0;long reads=0;mm_batch_t b;
  while(more){int32_t r=mmh_loader_next(ld,set,&b,&more);if(r<0)return 1;reads+=b.n_reads;set^=1;}
  double t1=now(),c1=cpu();
  printf("threads %d reads %ld wall %.3f s main-thread cpu %.3f s = %.2f us/read\n",threads,reads,t1-t0,c1-c0,(c1-c0)/reads*1e6);
  mmh_loader_close(ld);return 0;}
