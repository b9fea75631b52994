/* Host-side check of the filter-tree builder (ray-tracing_amd/csrc/rt_own_bvh.hpp), built and run by tests/test_host_sanitizers.py
 * with the address and undefined-behaviour sanitizers: for 1 to 200 000 random triangles (a third of them identical, which
 * sends ranges to the median fallback), an index subset, leaves of 1 / 2 / 4 and 1 / 3 / 8 host threads it checks that the
 * order is a permutation, that the leaves cover every primitive exactly once with at most `leaf` each, that children come
 * after their parents, and that the tree is the same for every thread count. No GPU is touched. */
#include <chrono>
#include <cstdio>
#include <random>
#include "rt_own_bvh.hpp"
static uint64_t hash_tree(const rtown::Built &b) {
  uint64_t h=1469598103934665603ull; for (auto &w: b.wide){ const unsigned char*p=(const unsigned char*)&w; for(size_t i=0;i<sizeof(w);i++){h^=p[i];h*=1099511628211ull;} }
  for (auto v: b.order){ h^=(uint32_t)v; h*=1099511628211ull; }
  return h;
}
int main(int argc,char**argv){
  std::mt19937_64 g(7);
  for (int64_t n : {1,2,3,4,5,7,64,1000,70000,200000}) for (int leaf : {1,2,4}) {
    std::vector<rt_tri> t((size_t)n); std::uniform_real_distribution<double> U(-30,30), S(-0.5,0.5);
    for (auto &x: t){ for(int a=0;a<3;a++){x.A[a]=U(g); x.e1[a]=S(g); x.e2[a]=S(g);} x.orig_id=0; x.mesh=0; }
    if (n >= 1000) for (int64_t k = 0; k < n/3; k++) t[(size_t)k] = t[0]; /* many equal centres: the median fallback */
    std::vector<int32_t> keep; for (int64_t k=0;k<n;k++) if (k%5!=3 || n<10) keep.push_back((int32_t)k);
    uint64_t h0=0;
    for (const char* thr : {"1","3","8"}) {
      setenv("OMP_NUM_THREADS", thr, 1);
      rtown::Built b = rtown::build(t.data(), keep.data(), (int64_t)keep.size(), leaf);
      if (!b.ok) { printf("n=%lld leaf=%d not ok\n",(long long)n,leaf); return 1; }
      // checks: order is a permutation; leaves cover everything once with counts <= leaf; child indices forward
      std::vector<char> seen(keep.size(),0); for (auto v: b.order){ if (v<0||(size_t)v>=keep.size()||seen[(size_t)v]) { printf("bad order\n"); return 1;} seen[(size_t)v]=1; }
      std::vector<char> cov(keep.size(),0); size_t covered=0;
      for (size_t r=0;r<b.wide.size();r++) for (int c=0;c<4;c++){ uint32_t code=b.wide[r].child[c]; if (code==RT_CHILD_EMPTY) continue; if (code>>31){ uint32_t first=code&0x0FFFFFFFu, cnt=((code>>28)&7u)+1u; if ((int)cnt>leaf|| first+cnt>keep.size()){printf("bad leaf n=%lld leaf=%d cnt=%u\n",(long long)n,leaf,cnt);return 1;} for(uint32_t k=first;k<first+cnt;k++){ if(cov[k]){printf("double cover\n");return 1;} cov[k]=1; covered++; } } else if (code<=r || code>=b.wide.size()){ printf("bad child\n"); return 1; } }
      if (covered!=keep.size()){ printf("n=%lld leaf=%d covered %zu of %zu\n",(long long)n,leaf,covered,keep.size()); return 1; }
      uint64_t h=hash_tree(b); if (!h0) h0=h; else if (h!=h0){ printf("n=%lld leaf=%d: tree depends on the thread count\n",(long long)n,leaf); return 1; }
    }
  }
  /* an exception on a worker thread or on the caller's own part reaches the caller, after every part has ended, and the
   * pool serves the next pass; the same through threads started for the pass */
  for (int pooled = 0; pooled < 2; pooled++) for (int thrower : {0, 2}) {
    std::unique_ptr<rtown::PoolScope> scope;
    if (pooled) scope.reset(new rtown::PoolScope(4));
    std::atomic<int> ran(0);
    bool caught = false;
    try { rtown::on_threads(4, [&](int p) { ran++; if (p == thrower) throw std::bad_alloc(); }); }
    catch (const std::bad_alloc &) { caught = true; }
    if (!caught || ran != 4) { printf("exception not carried over (pooled %d, thrower %d, ran %d)\n", pooled, thrower, (int)ran); return 1; }
    std::atomic<int> again(0);
    rtown::on_threads(4, [&](int) { again++; });
    if (again != 4) { printf("pool not usable after an exception\n"); return 1; }
  }
  printf("builder checks passed\n");
}
