// tools/replay_stats.cpp -- CPU replay of k_sf's decisions (filter -> probe -> resolve) over a synthetic text, using the
// product's own image and per-position code (am_image.h).  Analysis aid only: how many positions pass each stage, and how
// many dependent steps the trie walk of a deferred position takes.  Driven by tools/replay_stats.py.
#ifndef AM_REPLAY_CASE
#define AM_REPLAY_CASE 1
#endif
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>
#include "am_flatten.h"
using namespace am;
template <class T> std::vector<T> rd(const char* f){ FILE* fp=fopen(f,"rb"); fseek(fp,0,SEEK_END); long n=ftell(fp); fseek(fp,0,SEEK_SET); std::vector<T> v(n/sizeof(T)); if (fread(v.data(),1,n,fp) != (size_t)n) {} fclose(fp); return v; }
int main(){
  auto tr=rd<uint64_t>("tr.bin"); auto of=rd<uint32_t>("of.bin"); auto ra=rd<uint64_t>("ra.bin"); auto vl=rd<uint32_t>("vl.bin"); auto text=rd<uint8_t>("text.bin");
  RefArrays ref{tr.data(),tr.size(),of.data(),of.size()-1,ra.data(),vl.data()};
  std::vector<uint8_t> img; std::string err; if(flatten(ref,AM_REPLAY_CASE,img,err)){printf("err %s\n",err.c_str());return 1;}
  ImageHeader h; memcpy(&h,img.data(),sizeof(h)); SfView s=make_sf_view(img.data(),h);
  constexpr bool IC = AM_REPLAY_CASE != 0;
  uint64_t n=text.size(), cand=0, deferred=0, found=0, walked=0;
  uint64_t hist[32]; memset(hist,0,sizeof(hist));
  text.resize(n+64,0);
  for(uint64_t p=8;p<n;p++){
    uint32_t w=0; for(int i=0;i<4;i++) w|=(uint32_t)(IC ? fold_byte(text[p-3+i]) : text[p-3+i])<<(8*i);
    if(!sf_filter_window(s.bloom,s.bloom_log2_words,s.tiers,w)) continue;
    cand++;
    uint32_t w1,w2; load_suffix8(text.data(),p,w1,w2); if(IC){w1=fold_dword(w1);w2=fold_dword(w2);}
    const uint32_t wa[1]={w1}, nba[1]={(w2>>24)|(((w2>>16)&0xFFu)<<8)}; const uint64_t av[1]={p+1}; const bool vv[1]={true};
    bool defer[1]; uint32_t hint[1];
    sf_probe_n<1>(s,wa,nba,av,vv,defer,hint);
    if(!defer[0]) continue;
    deferred++;
    uint64_t it[8]={0,0,0,0,0,0,0,0};
    const uint64_t g[1]={p}; bool f[1]; uint32_t st[1],vl1[1];
    sf_resolve_n<IC,1>(s,text.data(),g,av,vv,hint,f,st,vl1,SfNoHook(),it);
    if(f[0]) found++;
    hist[it[0]<31?it[0]:31]++;
    if(it[0]) walked++;
  }
  printf("image: nodes %u edges %llu maps %llu, t4 buckets 2^%u, bloom 2^%u words\n", h.sf_n_nodes,(unsigned long long)h.n_edges,(unsigned long long)h.n_edge_maps,h.tier_log2_cap[3],h.sf_bloom_log2_words);
  printf("per KiB: candidates %.2f deferred %.2f found %.2f walked %.2f\n", cand*1024.0/n, deferred*1024.0/n, found*1024.0/n, walked*1024.0/n);
  printf("walk-loop iterations per deferred item:");
  double mean=0; for(int i=0;i<32;i++){ if(hist[i]) printf(" %d:%.1f%%",i,100.0*hist[i]/(deferred?deferred:1)); mean+=i*(double)hist[i]; }
  printf("  mean %.2f\n", mean/(deferred?deferred:1));
  return 0;
}
