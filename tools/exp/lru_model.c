/* lru_model.c — CPU model of lfx_match7.hip's formulation (round 5): two-level bucket LRU with exact tags + the resolver's walk over
 * duplicate-collapsed links, against the exact "most recent earlier occurrence of the 3-byte prefix inside the window"
 * (libflate_lz77/src/default.rs:76-87).  Prints, per kind of data and per candidate key mix, the share of positions whose head
 * entry carries another tag, the share left unresolved after two levels, the resolver's hops — and WRONG must be 0.
 *   gcc -O2 -o lru_model tools/exp/lru_model.c tools/synth.c -lm && ./lru_model 0 [MiB [mix]]   (0 text, 1 lowent, 2 random,
 *   3 "abc", 4 nibbles; mix 6 = the kernel's multiplier; exit status 1 when any answer is wrong)
 * Not product code, not the oracle: a design worksheet. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
void lfx_synth_text(void*, size_t, uint64_t);
void lfx_synth_lowent(void*, size_t, uint64_t);
#define W 32768
#define BITS 14
#define TAGB (24-BITS)
static uint32_t mix(uint32_t k,int mode){
  switch(mode){
   case 0: return (k*0x9E3779B1u)&0xFFFFFF;
   case 1: return (k*2654435761u)&0xFFFFFF;            // low 24 bits of product (bijective, odd multiplier)
   case 2: { uint32_t x=(k*0x9E3779B1u)&0xFFFFFF; x^=x>>12; return (x*0x85EBCA6Bu)&0xFFFFFF; }
   case 3: { uint32_t x=k; x^=x>>9; x=(x*0x2C1B3C6Du)&0xFFFFFF; x^=x>>13; x=(x*0x297A2D39u)&0xFFFFFF; x^=x>>11; return x; }
   case 4: { uint32_t x=(k*0x00C5A3B5u)&0xFFFFFF; return x; }
   case 5: { uint32_t x=k^(k>>7); x=(x*0x9E3779B1u)&0xFFFFFF; return x; }
   case 6: return (k*0x00374ADDu)&0xFFFFFF;           // the kernel's KEY_MULT (tools/exp/lru_mult.c picked it)
  }
  return k;
}
int main(int argc,char**argv){
  size_t N = (size_t)(argc>2?atoi(argv[2]):16)<<20; int kind = argc>1?atoi(argv[1]):0;
  const int only = argc>3?atoi(argv[3]):-1;          /* one key mix instead of all seven */
  uint64_t wrong_all=0;
  uint8_t*buf=malloc(N+8);
  if(kind==1) lfx_synth_lowent(buf,N,0x5EED0005); else if(kind==0) lfx_synth_text(buf,N,0x5EED0002);
  else if(kind==2){ uint64_t s=88172645463325252ull; for(size_t i=0;i<N;i++){ s^=s<<13;s^=s>>7;s^=s<<17; buf[i]=s>>32; } }
  else if(kind==3){ uint64_t s=88172645463325252ull; for(size_t i=0;i<N;i++){ s^=s<<13;s^=s>>7;s^=s<<17; buf[i]="abc"[(s>>32)%3]; } }
  else if(kind==4){ uint64_t s=88172645463325252ull; for(size_t i=0;i<N;i++){ s^=s<<13;s^=s>>7;s^=s<<17; buf[i]=(s>>32)&15; } }
  size_t CH=262144;
  int32_t *last = malloc(sizeof(int32_t)<<24);
  uint32_t *head=malloc(4<<BITS), *sec=malloc(4<<BITS);
  uint16_t *cd=malloc(2*CH), *cl=malloc(2*CH);
  for(int mode=0;mode<7;mode++){
  if(only>=0&&mode!=only) continue;
  uint64_t nAll=0, mism=0, unres=0, wrong=0, hops=0, maxh=0;
  for(size_t c0=0;c0+CH<=N;c0+=CH){
    uint8_t*b=buf+c0; size_t n=CH; size_t end=n-3;
    for(size_t i=0;i<(1u<<24);i++) last[i]=-1;
    memset(head,0,4<<BITS); memset(sec,0,4<<BITS);
    for(size_t p=0;p<end;p++){
      uint32_t k=b[p]|b[p+1]<<8|b[p+2]<<16;
      uint32_t kk=mix(k,mode); uint32_t idx=kk>>TAGB, tag=kk&((1u<<TAGB)-1);
      uint32_t spos=p+32769; uint32_t ent=spos<<TAGB|tag;
      uint32_t o1=head[idx]; head[idx]=ent;
      int same1=(o1&((1u<<TAGB)-1))==tag; uint32_t d1=spos-(o1>>TAGB);
      uint32_t s; if(!same1){ s=sec[idx]; sec[idx]=o1; } else s=sec[idx];
      int same2=(s&((1u<<TAGB)-1))==tag; uint32_t d2=spos-(s>>TAGB);
      uint32_t c,l;
      if(same1){ c=d1<=W?d1:0; l=d2<=W?d2:0; }
      else { mism++; l=d1<=W?d1:0; if(d1>W||d2>W) c=0; else if(same2) c=d2; else c=0xFFFF; }
      cd[p]=c; cl[p]=l; nAll++;
    }
    // K_c
    for(size_t p=0;p<end;p++){
      uint32_t k=b[p]|b[p+1]<<8|b[p+2]<<16;
      if(cd[p]==0xFFFF){ unres++; uint32_t r=p-cl[p]; uint32_t h=0; uint32_t c=0;
        for(;;){ uint32_t l=cl[r]; if(l==0||p-(r-l)>W){c=0;break;} r-=l; h++; uint32_t kr=b[r]|b[r+1]<<8|b[r+2]<<16; if(kr==k){c=p-r;break;} }
        cd[p]=c; hops+=h; if(h>maxh)maxh=h; }
      int32_t q=last[k]; uint32_t ex=(q>=0&&p-q<=W)?p-q:0; last[k]=p;
      if(cd[p]!=ex) wrong++;
    }
  }
  printf("kind=%d mode=%d mismatch=%.4f unresolved=%.4f hops/unres=%.2f maxhops=%lu WRONG=%lu\n",kind,mode,(double)mism/nAll,(double)unres/nAll,unres?(double)hops/unres:0.0,maxh,wrong);
  wrong_all+=wrong;
  }
  return wrong_all!=0;
}
