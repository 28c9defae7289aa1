/* Exhaustive proof used by the device code (pathtracer_amd/csrc/pt_device.hpp, diffuse_reflection):
 * for EVERY xorshift32 state, the three draws that follow never satisfy the early-exit condition
 * of the reference's GetDiffuseReflection loop (PT_CudaSimple/Functions.cuh:99-104),
 *     ((x*x + y*y) + z*z) + 1.0f <= 1.0f   with x,y,z = 2u-1,
 * so the loop always runs its full 11 rounds (33 draws) and only the 11th sample is used.
 * Prints "early exits 0" and returns 0 when that holds.  ~1 s on 8 cores. */
#include <stdio.h>
#include <stdint.h>
static inline uint32_t xs(uint32_t x){x^=x<<13;x^=x>>17;x^=x<<5;return x;}
int main(){
  uint64_t found=0, filt=0;
  #pragma omp parallel for reduction(+:found,filt) schedule(static)
  for(uint64_t s=1;s<=0xFFFFFFFFull;s++){
    uint32_t x0=xs((uint32_t)s);
    if((uint32_t)(x0-0x7FF00000u)>0x00200000u) continue;
    filt++;
    uint32_t x1=xs(x0),x2=xs(x1);
    float ax=(float)x0/4294967296.0f*2.0f-1.0f, ay=(float)x1/4294967296.0f*2.0f-1.0f, az=(float)x2/4294967296.0f*2.0f-1.0f;
    float len2=ax*ax+ay*ay+az*az+1.0f;
    if(!(len2>1.0f)){found++; printf("state %u -> %u %u %u len2 %.9g\n",(uint32_t)s,x0,x1,x2,len2);}
  }
  printf("filter passes %llu, early exits %llu\n",(unsigned long long)filt,(unsigned long long)found);
  return found!=0;
}
