#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* Dout) {
  // A: 16x4 col-major (ld 16), B: 4x16 row... give raw per-lane operands: a[l], b[l]
  int l = threadIdx.x;
  double a = A[l], b = B[l];
  v4d c = {0,0,0,0};
  v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) Dout[l*4 + r] = d[r];
}
int main() {
  double hA[64], hB[64], hD[256];
  for (int i = 0; i < 64; i++) { hA[i] = 1 + (rand()%97)*0.01; hB[i] = 2 + (rand()%89)*0.01; }
  double *dA,*dB,*dD; hipMalloc(&dA,512); hipMalloc(&dB,512); hipMalloc(&dD,2048);
  hipMemcpy(dA,hA,512,hipMemcpyHostToDevice); hipMemcpy(dB,hB,512,hipMemcpyHostToDevice);
  k<<<1,64>>>(dA,dB,dD); hipMemcpy(hD,dD,2048,hipMemcpyDeviceToHost);
  // hypothesis: A(i,k) in lane i+16k ; B(k,j) in lane j+16k ; test D layouts
  auto Aik=[&](int i,int kk){return hA[i+16*kk];}; auto Bkj=[&](int kk,int j){return hB[j+16*kk];};
  double e1=0,e2=0;
  for (int l=0;l<64;l++) for(int r=0;r<4;r++){
    int j=l%16; int i1=4*(l/16)+r; int i2=(l/16)+4*r;
    double s1=0,s2=0; for(int kk=0;kk<4;kk++){ s1+=Aik(i1,kk)*Bkj(kk,j); s2+=Aik(i2,kk)*Bkj(kk,j);} 
    e1=fmax(e1,fabs(hD[l*4+r]-s1)); e2=fmax(e2,fabs(hD[l*4+r]-s2)); }
  printf("layout H1 (i = 4*(lane/16)+r): max err %g\nlayout H2 (i = lane/16 + 4*r): max err %g\n", e1, e2);
  // summation order check: does MFMA sum k ascending with fma chain? compare bitwise to fma chain
  int exact=0; for (int l=0;l<64;l++) for(int r=0;r<4;r++){ int j=l%16; int i=(e1<e2)?4*(l/16)+r:(l/16)+4*r; double s=0; for(int kk=0;kk<4;kk++) s=fma(Aik(i,kk),Bkj(kk,j),s); exact += (s==hD[l*4+r]); }
  printf("bitwise equal to ascending fma chain: %d / 256\n", exact);
  return 0; }
