// Host-side finish of an MSM (g1_host64.hpp) priced on the host CPU: one multiplication, one inversion, and a chain shaped like
// msm_finish's (20 plane points converted, ~21 additions, 18 doublings).  g++ / clang++ -O3 -std=c++17 -I. tools/host_finish_bench.cpp
#include "typlonk_amd/csrc/g1_host64.hpp"
#include <chrono>
#include <cstdio>
using namespace ty::h64;
int main(){
  Fq a={{1,2,3,4,5,6}}, b={{7,8,9,10,11,0x1}};
  auto t0=std::chrono::steady_clock::now();
  for(int i=0;i<1000000;i++){a=mul(a,b);}
  auto t1=std::chrono::steady_clock::now();
  printf("mul %.1f ns (%llx)\n", std::chrono::duration<double,std::nano>(t1-t0).count()/1e6,(unsigned long long)a.v[0]);
  t0=std::chrono::steady_clock::now();
  for(int i=0;i<1000;i++){a=inv(a); a.v[0]^=1;}
  t1=std::chrono::steady_clock::now();
  printf("inv %.1f ns (%llx)\n", std::chrono::duration<double,std::nano>(t1-t0).count()/1e3,(unsigned long long)a.v[0]);
  // a chain like the finish: 20 from_device, 21 adds, 18 dbl (on garbage points: same arithmetic)
  Xyzz p; p.x=a;p.y=b;p.zz=mul(a,b);p.zzz=mul(p.zz,b);
  t0=std::chrono::steady_clock::now();
  Xyzz acc=p;
  for(int r=0;r<1000;r++){
    for(int i=0;i<21;i++){acc=xyzz_add(acc,p); p.x.v[0]^=acc.x.v[1]&0xff;}
    for(int i=0;i<18;i++)acc=xyzz_dbl(acc);
    for(int i=0;i<80;i++){a=mul(a,b);} // 20 points x 4 conversions
  }
  t1=std::chrono::steady_clock::now();
  printf("finish chain %.1f us (%llx)\n", std::chrono::duration<double,std::micro>(t1-t0).count()/1e3,(unsigned long long)(acc.x.v[0]^a.v[0]));
}
