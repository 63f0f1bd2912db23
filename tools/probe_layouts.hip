// Probe of gfx950 fragment layouts used by the kernels in mca-paper_amd/csrc.
// Exact small-integer data; every check prints PASS/FAIL.  Build:
//   hipcc --offload-arch=gfx950 -O2 tools/probe_layouts.hip -o gpurun_out/probe && gpurun_out/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

__device__ __host__ inline unsigned short f2bf(float f){ unsigned u; memcpy(&u,&f,4); return (unsigned short)((u + 0x7FFF + ((u>>16)&1))>>16); }
__device__ __host__ inline float bf2f(unsigned short b){ unsigned u=((unsigned)b)<<16; float f; memcpy(&f,&u,4); return f; }

// ---- test 1: 32x32x16 bf16.  A[32][16], B[16][32] row-major in global. C[32][32].
__global__ void k_mfma32(const unsigned short* A, const unsigned short* B, float* C){
  int l = threadIdx.x, r = l&31, h = l>>5;
  bf16x8 a, b;
  for(int j=0;j<8;j++){ a[j] = A[r*16 + 8*h + j]; b[j] = B[(8*h+j)*32 + r]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,c,0,0,0);
  for(int i=0;i<16;i++){ int row = (i&3) + 8*(i>>2) + 4*h; C[row*32 + r] = c[i]; }
}
// ---- test 2: 16x16x32 bf16. A[16][32], B[32][16]. C[16][16]
__global__ void k_mfma16(const unsigned short* A, const unsigned short* B, float* C){
  int l = threadIdx.x, r = l&15, q = l>>4;
  bf16x8 a, b;
  for(int j=0;j<8;j++){ a[j] = A[r*32 + 8*q + j]; b[j] = B[(8*q+j)*16 + r]; }
  f32x4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,c,0,0,0);
  for(int i=0;i<4;i++){ int row = q*4 + i; C[row*16 + r] = c[i]; }
}
// ---- test 3: accumulator as B operand.  X = A1[32][16]*B1[16][32] (32x32), Y = A2[32][32] * X (32x32)
__global__ void k_acc_as_b(const unsigned short* A1, const unsigned short* B1, const unsigned short* A2, float* Y){
  int l = threadIdx.x, r = l&31, h = l>>5;
  bf16x8 a, b;
  for(int j=0;j<8;j++){ a[j] = A1[r*16 + 8*h + j]; b[j] = B1[(8*h+j)*32 + r]; }
  f32x16 x = {0};
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,x,0,0,0);
  f32x16 y = {0};
  for(int s=0;s<2;s++){
    bf16x8 xb, a2;
    for(int j=0;j<8;j++){
      xb[j] = (short)f2bf(x[8*s+j]);
      int k = 16*s + 8*(j>>2) + 4*h + (j&3);   // row of X carried by element j
      a2[j] = A2[r*32 + k];
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, y, 0,0,0);
  }
  for(int i=0;i<16;i++){ int row = (i&3) + 8*(i>>2) + 4*h; Y[row*32 + r] = y[i]; }
}
// ---- test 3b: accumulator as A operand: Z = X^T * B2, X 32x32, B2[32][32]
__global__ void k_acc_as_a(const unsigned short* A1, const unsigned short* B1, const unsigned short* B2, float* Z){
  int l = threadIdx.x, r = l&31, h = l>>5;
  bf16x8 a, b;
  for(int j=0;j<8;j++){ a[j] = A1[r*16 + 8*h + j]; b[j] = B1[(8*h+j)*32 + r]; }
  f32x16 x = {0};
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,x,0,0,0);
  f32x16 z = {0};
  for(int s=0;s<2;s++){
    bf16x8 xa, b2;
    for(int j=0;j<8;j++){
      xa[j] = (short)f2bf(x[8*s+j]);
      int k = 16*s + 8*(j>>2) + 4*h + (j&3);
      b2[j] = B2[k*32 + r];
    }
    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, b2, z, 0,0,0);
  }
  for(int i=0;i<16;i++){ int row = (i&3) + 8*(i>>2) + 4*h; Z[row*32 + r] = z[i]; }
}
// ---- test 4: ds_read_b64_tr_b16.  LDS tile T[16 rows][64 cols] u16, value = row*64+col.
// lane l: group g=l>>4, i=l&15, q=i>>2, p=i&3. address -> row (4*(g&1)+q)... we just record raw results for
// a simple addressing: lane supplies &T[rowbase + q][colbase + 4p], rowbase = 4*(g>>1), colbase=16*(g&1)
__global__ void k_tr(unsigned short* out){
  __shared__ unsigned short T[16*64];
  for(int i=threadIdx.x;i<16*64;i+=64) T[i] = (unsigned short)i;
  __syncthreads();
  int l = threadIdx.x, g=l>>4, i=l&15, q=i>>2, p=i&3;
  int row = 4*(g>>1) + q, col = 16*(g&1) + 4*p;
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(&T[row*64+col]));
  for(int j=0;j<4;j++) out[l*4+j] = (unsigned short)v[j];
}
// ---- test 5: permlane32_swap
__global__ void k_swap(unsigned* out){
  unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
  auto s = __builtin_amdgcn_permlane32_swap(a,b,false,false);
  out[threadIdx.x*2] = s[0]; out[threadIdx.x*2+1] = s[1];
}
// ---- test 6: V^T A-operand through tr reads, full product O^T[d][q] = sum_key V[key][d] * P^T[key][q]
// V tile [32 keys][32 d] row-major in LDS (64B rows), X (=P^T [key][q]) from an MFMA.
__global__ void k_pv(const unsigned short* A1, const unsigned short* B1, const unsigned short* V, float* O){
  __shared__ unsigned short Vt[32*32];
  for(int i=threadIdx.x;i<32*32;i+=64) Vt[i] = V[i];
  __syncthreads();
  int l = threadIdx.x, r = l&31, h = l>>5;
  bf16x8 a, b;
  for(int j=0;j<8;j++){ a[j] = A1[r*16 + 8*h + j]; b[j] = B1[(8*h+j)*32 + r]; }
  f32x16 x = {0};
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,x,0,0,0);   // X[key][q]
  f32x16 o = {0};
  for(int s=0;s<2;s++){
    bf16x8 xb, va;
    for(int j=0;j<8;j++) xb[j] = (short)f2bf(x[8*s+j]);
    // A operand: row = d = l&31, k element j <-> key 16s + 8(j>>2) + 4h + (j&3)
    for(int t=0;t<2;t++){
      int key = 16*s + 8*t + 4*h + ((l&15)>>2);
      int d   = 16*((l>>4)&1) + 4*(l&3);
      v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(&Vt[key*32 + d]));
      for(int j=0;j<4;j++) va[4*t+j] = v[j];
    }
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, xb, o, 0,0,0);
  }
  for(int i=0;i<16;i++){ int row = (i&3) + 8*(i>>2) + 4*h; O[row*32 + r] = o[i]; }   // O^T[d=row][q=r]
}

static int check(const char* name, const std::vector<float>& got, const std::vector<float>& ref){
  int bad=0; for(size_t i=0;i<ref.size();i++) if(got[i]!=ref[i]) { if(bad<4) printf("  %s mismatch at %zu: got %g ref %g\n",name,i,got[i],ref[i]); bad++; }
  printf("%s: %s (%d bad of %zu)\n", name, bad?"FAIL":"PASS", bad, ref.size()); return bad;
}

int main(){
  srand(1);
  auto rnd=[&](){ return (float)((rand()%7)-3); };
  std::vector<unsigned short> A(32*16), B(16*32), A2(32*32), B2(32*32), A16(16*32), B16(32*16), V(32*32);
  std::vector<float> Af(32*16), Bf(16*32), A2f(32*32), B2f(32*32), A16f(16*32), B16f(32*16), Vf(32*32);
  for(int i=0;i<32*16;i++){ Af[i]=rnd(); A[i]=f2bf(Af[i]); Bf[i]=rnd(); B[i]=f2bf(Bf[i]); A16f[i]=rnd(); A16[i]=f2bf(A16f[i]); B16f[i]=rnd(); B16[i]=f2bf(B16f[i]); }
  for(int i=0;i<32*32;i++){ A2f[i]=rnd(); A2[i]=f2bf(A2f[i]); B2f[i]=rnd(); B2[i]=f2bf(B2f[i]); Vf[i]=rnd(); V[i]=f2bf(Vf[i]); }
  unsigned short *dA,*dB,*dA2,*dB2,*dA16,*dB16,*dV,*dU; float* dC; unsigned* dW;
  CK(hipMalloc(&dA,A.size()*2)); CK(hipMalloc(&dB,B.size()*2)); CK(hipMalloc(&dA2,A2.size()*2)); CK(hipMalloc(&dB2,B2.size()*2));
  CK(hipMalloc(&dA16,A16.size()*2)); CK(hipMalloc(&dB16,B16.size()*2)); CK(hipMalloc(&dV,V.size()*2));
  CK(hipMalloc(&dC,32*32*4)); CK(hipMalloc(&dU,64*4*2)); CK(hipMalloc(&dW,64*2*4));
  CK(hipMemcpy(dA,A.data(),A.size()*2,hipMemcpyHostToDevice)); CK(hipMemcpy(dB,B.data(),B.size()*2,hipMemcpyHostToDevice));
  CK(hipMemcpy(dA2,A2.data(),A2.size()*2,hipMemcpyHostToDevice)); CK(hipMemcpy(dB2,B2.data(),B2.size()*2,hipMemcpyHostToDevice));
  CK(hipMemcpy(dA16,A16.data(),A16.size()*2,hipMemcpyHostToDevice)); CK(hipMemcpy(dB16,B16.data(),B16.size()*2,hipMemcpyHostToDevice));
  CK(hipMemcpy(dV,V.data(),V.size()*2,hipMemcpyHostToDevice));
  int fails=0;
  std::vector<float> got(32*32), ref(32*32), X(32*32);
  // 1
  for(int i=0;i<32;i++)for(int j=0;j<32;j++){ float s=0; for(int k=0;k<16;k++) s+=Af[i*16+k]*Bf[k*32+j]; X[i*32+j]=s; }
  hipLaunchKernelGGL(k_mfma32,1,64,0,0,dA,dB,dC); CK(hipMemcpy(got.data(),dC,32*32*4,hipMemcpyDeviceToHost));
  fails+=check("mfma_32x32x16_bf16 layout",got,X);
  // 2
  { std::vector<float> g(256), r(256);
    for(int i=0;i<16;i++)for(int j=0;j<16;j++){ float s=0; for(int k=0;k<32;k++) s+=A16f[i*32+k]*B16f[k*16+j]; r[i*16+j]=s; }
    hipLaunchKernelGGL(k_mfma16,1,64,0,0,dA16,dB16,dC); CK(hipMemcpy(g.data(),dC,256*4,hipMemcpyDeviceToHost));
    fails+=check("mfma_16x16x32_bf16 layout",g,r); }
  // 3
  for(int i=0;i<32;i++)for(int j=0;j<32;j++){ float s=0; for(int k=0;k<32;k++) s+=A2f[i*32+k]*X[k*32+j]; ref[i*32+j]=s; }
  hipLaunchKernelGGL(k_acc_as_b,1,64,0,0,dA,dB,dA2,dC); CK(hipMemcpy(got.data(),dC,32*32*4,hipMemcpyDeviceToHost));
  fails+=check("acc as B operand (Y=A*X)",got,ref);
  for(int i=0;i<32;i++)for(int j=0;j<32;j++){ float s=0; for(int k=0;k<32;k++) s+=X[k*32+i]*B2f[k*32+j]; ref[i*32+j]=s; }
  hipLaunchKernelGGL(k_acc_as_a,1,64,0,0,dA,dB,dB2,dC); CK(hipMemcpy(got.data(),dC,32*32*4,hipMemcpyDeviceToHost));
  fails+=check("acc as A operand (Z=X^T*B)",got,ref);
  // 4
  { std::vector<unsigned short> u(256); hipLaunchKernelGGL(k_tr,1,64,0,0,dU); CK(hipMemcpy(u.data(),dU,512,hipMemcpyDeviceToHost));
    int bad=0; for(int l=0;l<64;l++){ int g=l>>4,i=l&15; for(int j=0;j<4;j++){ int row=4*(g>>1)+j, col=16*(g&1)+i; int exp=row*64+col; if(u[l*4+j]!=exp){ if(bad<6) printf("  tr lane %d elem %d got (r%d,c%d) expect (r%d,c%d)\n",l,j,u[l*4+j]/64,u[l*4+j]%64,row,col); bad++; } } }
    printf("ds_read_b64_tr_b16 semantics: %s\n", bad?"FAIL":"PASS"); fails+=bad; }
  // 5
  { std::vector<unsigned> w(128); hipLaunchKernelGGL(k_swap,1,64,0,0,dW); CK(hipMemcpy(w.data(),dW,512,hipMemcpyDeviceToHost));
    // expectation: lanes 0-31: s0 = a(own), s1 = a of lane+32 ; lanes 32-63: s0 = b of lane-32, s1 = b own
    int bad=0; for(int l=0;l<64;l++){ unsigned e0 = l<32? (unsigned)l : (unsigned)(1000+l-32); unsigned e1 = l<32? (unsigned)(l+32) : (unsigned)(1000+l); if(w[2*l]!=e0||w[2*l+1]!=e1){ if(bad<6) printf("  swap lane %d got (%u,%u) expect (%u,%u)\n",l,w[2*l],w[2*l+1],e0,e1); bad++; } }
    printf("permlane32_swap semantics: %s\n", bad?"FAIL":"PASS"); fails+=bad; }
  // 6
  for(int d=0;d<32;d++)for(int q=0;q<32;q++){ float s=0; for(int k=0;k<32;k++) s+=Vf[k*32+d]*X[k*32+q]; ref[d*32+q]=s; }
  hipLaunchKernelGGL(k_pv,1,64,0,0,dA,dB,dV,dC); CK(hipMemcpy(got.data(),dC,32*32*4,hipMemcpyDeviceToHost));
  fails+=check("O^T = V^T (tr reads) * P^T (acc as B)",got,ref);
  printf("TOTAL %s\n", fails?"FAIL":"PASS");
  return fails?1:0;
}
