/* lru_mult.c — the unresolved share of lfx_match7.hip's two-level LRU for a list of odd 24-bit multipliers, on two seeds of the
 * TEXT generator, a file `real.txt` in the working directory (any text) and LOWENT (how 0x374ADD was chosen).
 *   gcc -O2 -o lru_mult tools/exp/lru_mult.c tools/synth.c -lm && ./lru_mult */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
void lfx_synth_text(void*, size_t, uint64_t);
void lfx_synth_lowent(void*, size_t, uint64_t);
#define W 32768
#define BITS 14
#define TAGB 10
static double run(uint8_t*buf,size_t N,uint32_t M,double*mm){
  size_t CH=262144; static uint32_t head[1<<BITS], sec[1<<BITS];
  uint64_t nAll=0, unres=0, mism=0;
  for(size_t c0=0;c0+CH<=N;c0+=CH){
    uint8_t*b=buf+c0; size_t end=CH-3;
    memset(head,0,sizeof head); memset(sec,0,sizeof sec);
    for(size_t p=0;p<end;p++){
      uint32_t k=b[p]|b[p+1]<<8|b[p+2]<<16;
      uint32_t kk=(k*M)&0xFFFFFF; uint32_t idx=kk>>TAGB, tag=kk&1023;
      uint32_t spos=p+32769; uint32_t ent=spos<<TAGB|tag;
      uint32_t o1=head[idx]; head[idx]=ent;
      int same1=(o1&1023)==tag; uint32_t d1=spos-(o1>>TAGB);
      uint32_t s2=sec[idx]; if(!same1) sec[idx]=o1;
      int same2=(s2&1023)==tag; uint32_t d2=spos-(s2>>TAGB);
      if(!same1){ if(d1<=W)mism++; if(d1<=W&&d2<=W&&!same2) unres++; }
      nAll++;
    }
  }
  *mm=(double)mism/nAll; return (double)unres/nAll;
}
int main(){
  size_t N=8u<<20; uint8_t *t1=malloc(N+8),*t2=malloc(N+8),*t3=malloc(N+8),*t4=malloc(N+8);
  lfx_synth_text(t1,N,0x5EED0002); lfx_synth_text(t2,N,0x1234567); lfx_synth_lowent(t4,N,0x5EED0005);
  FILE*f=fopen("real.txt","rb"); size_t n3=0; if(f){ n3=fread(t3,1,N,f); fclose(f);} if(n3<262144){ memcpy(t3,t2,N); n3=N; }   /* (no file: the second seed again) */
  uint32_t Ms[]={0xC5A3B5,0x374ADD,0x8D4F45,0xD0578F,0xBA126D,0x3D3CDB,0x9E3779B1&0xFFFFFF,0x2C1B3C6D&0xFFFFFF|1};
  for(int i=0;i<8;i++){ double m; printf("M=0x%06X synth=%.5f synth2=%.5f", Ms[i], run(t1,N,Ms[i],&m), run(t2,N,Ms[i],&m)); printf(" real=%.5f", run(t3,n3,Ms[i],&m)); printf(" (mism %.3f) lowent=%.5f\n", m, run(t4,N,Ms[i],&m)); }
}
