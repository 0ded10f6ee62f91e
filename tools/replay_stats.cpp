// tools/replay_stats.cpp -- CPU replay of k_sf's decisions (filter -> probe -> resolve) over a synthetic text, using the
// product's own image and per-position code (am_image.h).  Analysis aid only: it counts how many positions pass each
// stage and why candidates reach phase 2.  Driven by tools/replay_stats.py (which dumps tr.bin/of.bin/ra.bin/vl.bin/text.bin).
#ifndef AM_REPLAY_CASE
#define AM_REPLAY_CASE 1
#endif
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>
#include "am_flatten.h"
using namespace am;
template <class T> std::vector<T> rd(const char* f){ FILE* fp=fopen(f,"rb"); fseek(fp,0,SEEK_END); long n=ftell(fp); fseek(fp,0,SEEK_SET); std::vector<T> v(n/sizeof(T)); fread(v.data(),1,n,fp); fclose(fp); return v; }
int main(){
  auto tr=rd<uint64_t>("tr.bin"); auto of=rd<uint32_t>("of.bin"); auto ra=rd<uint64_t>("ra.bin"); auto vl=rd<uint32_t>("vl.bin"); auto text=rd<uint8_t>("text.bin");
  RefArrays ref{tr.data(),tr.size(),of.data(),of.size()-1,ra.data(),vl.data()};
  std::vector<uint8_t> img; std::string err; if(flatten(ref,AM_REPLAY_CASE,img,err)){printf("err %s\n",err.c_str());return 1;}
  ImageHeader h; memcpy(&h,img.data(),sizeof(h)); SfView s=make_sf_view(img.data(),h);
  const uint32_t lb=s.tier_log2_cap[3];
  uint64_t n=text.size(), cand=0, deferred=0, found=0; uint64_t byk[4]={0,0,0,0}, fbyk[4]={0,0,0,0}; uint64_t keyhit=0; uint64_t nkeys=0; for (uint32_t i=0;i<(1u<<lb);i++){ if(s.t4_cold[i].z!=kNone) nkeys++; if(s.t4_cold[i].w!=kNone) nkeys++; }
  text.resize(n+64,0);
  for(uint64_t p=8;p<n;p++){
    uint32_t w=0; for(int i=0;i<4;i++) w|=(uint32_t)(AM_REPLAY_CASE ? fold_byte(text[p-3+i]) : text[p-3+i])<<(8*i);
    const uint32_t hh=bloom_hash(w,4);
    if(!bloom_hit(s.bloom[bloom_word(hh,s.bloom_log2_words)],hh)) continue;
    cand++;
    { u32x4 ca=s.t4_cold[t4_bucket(t4_hash_a(w),lb)], cb=s.t4_cold[t4_bucket(t4_hash_b(w),lb)]; if((ca.x==w&&ca.z!=kNone)||(ca.y==w&&ca.w!=kNone)||(cb.x==w&&cb.z!=kNone)||(cb.y==w&&cb.w!=kNone)) keyhit++; }
    uint32_t nb=AM_REPLAY_CASE ? fold_byte(text[p-4]) : text[p-4], nb2=AM_REPLAY_CASE ? fold_byte(text[p-5]) : text[p-5];
    uint32_t ha=t4_hash_a(w), hb=t4_hash_b(w), fp=t4_fingerprint(ha,lb), e=t4_expect(fp,nb|nb2<<8);
    u32x2 ba=s.t4_hot[t4_bucket(ha,lb)], bb=s.t4_hot[t4_bucket(hb,lb)];
    uint32_t sl[4]={ba.x,ba.y,bb.x,bb.y}; int hitk=-1;
    for(int i=0;i<4;i++) if(t4_slot_diff(sl[i],e)==0){ hitk=(sl[i]&31)/8; }
    if(hitk<0) continue;
    deferred++; byk[hitk]++;
    uint32_t st,vlen; bool f=sf_resolve<AM_REPLAY_CASE != 0>(s,text.data(),p,p+1,st,vlen);
    if(f) found++; else fbyk[hitk]++;
  }
  uint64_t setbits=0; for(uint32_t i=0;i<(1u<<s.bloom_log2_words);i++) setbits+=__builtin_popcount(s.bloom[i]);
  printf("slots used %llu of %u (buckets 2^%u), bloom fill %.3f, key-present candidates %.2f/KiB, bloom-FP candidates %.2f/KiB\n",(unsigned long long)nkeys,2u<<lb,lb,setbits/(32.0*(1u<<s.bloom_log2_words)),keyhit*1024.0/n,(cand-keyhit)*1024.0/n);
  printf("per KiB: cand %.2f deferred %.2f found %.2f | deferred by k8 0/8/16: %.2f %.2f %.2f | false by k8: %.2f %.2f %.2f\n", cand*1024.0/n, deferred*1024.0/n, found*1024.0/n,
     byk[0]*1024.0/n, byk[1]*1024.0/n, byk[2]*1024.0/n, fbyk[0]*1024.0/n, fbyk[1]*1024.0/n, fbyk[2]*1024.0/n);
}
