/* Mutated JPEG files through the texture decoder (ray-tracing_amd/host/rt_jpeg.cpp), built with the address and
 * undefined-behaviour sanitizers by tests/test_host_sanitizers.py: truncations, bit flips, random header bytes and stray
 * markers. A file is decoded or refused with an exception; anything else (a crash, a sanitizer report, a picture of the
 * wrong size) fails. Usage: jpeg_fuzz file.jpg ... (ITERS in the environment: mutations per file). With EXPECT_REFUSED set the
 * files are fixed cases that the decoder must refuse as they are (a header that would make it read outside its planes). */
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <random>
#include <string>
#include <vector>
#include <stdexcept>
namespace rt { void decodeJPEG(const uint8_t *data, size_t n, std::vector<uint8_t> &rgb, int &W, int &H); bool looksLikeJPEG(const uint8_t *data, size_t n); }
static std::vector<uint8_t> slurp(const char *p){ FILE*f=fopen(p,"rb"); std::vector<uint8_t> b; if(!f) return b; fseek(f,0,SEEK_END); long n=ftell(f); fseek(f,0,SEEK_SET); b.resize((size_t)n); if(fread(b.data(),1,(size_t)n,f)!=(size_t)n) b.clear(); fclose(f); return b; }
int main(int argc,char**argv){
  std::mt19937 g(1); long ok=0, bad=0, thrown=0;
  if (getenv("EXPECT_REFUSED")) {
    for (int a=1;a<argc;a++){
      std::vector<uint8_t> src=slurp(argv[a]); std::vector<uint8_t> rgb; int W=0,H=0;
      try { rt::decodeJPEG(src.data(), src.size(), rgb, W, H); printf("%s: decoded, should have been refused\n", argv[a]); return 1; }
      catch (const std::exception &e) { printf("%s: refused (%s)\n", argv[a], e.what()); }
    }
    return 0;
  }
  for (int a=1;a<argc;a++){
    std::vector<uint8_t> src=slurp(argv[a]); if(src.empty()) continue;
    for (int it=0; it<(int)(atoi(getenv("ITERS")?getenv("ITERS"):"40")); it++){
      std::vector<uint8_t> m=src;
      int kind=it%4;
      if (it==0) {} else if (kind==0){ m.resize(g()%m.size()); }
      else if (kind==1){ for(int k=0;k<1+(int)(g()%8);k++) m[g()%m.size()]^=(uint8_t)(1u<<(g()%8)); }
      else if (kind==2){ size_t p=g()%std::min<size_t>(m.size(),2048); for(int k=0;k<4 && p+k<m.size();k++) m[p+k]=(uint8_t)g(); }   /* headers */
      else { size_t p=g()%m.size(); m[p]=0xFF; if(p+1<m.size()) m[p+1]=(uint8_t)(0xC0+g()%0x3F); }               /* stray markers */
      std::vector<uint8_t> rgb; int W=0,H=0;
      try { rt::decodeJPEG(m.data(), m.size(), rgb, W, H); ok++; if (rgb.size()!=(size_t)W*H*3) { printf("size mismatch\n"); return 1; } }
      catch (const std::exception &e) { thrown++; }
    }
  }
  printf("decoded %ld, refused %ld, exceptions %ld\n", ok, bad, thrown);
}
