#include <hip/hip_runtime.h>
#include <cstdio>
// probe: lane layout of v_mfma_f64_4x4x4_4b_f64.  A(i,k)=10*i+k+100*b, B(k,j)= (k==j) -> D = A (if B is identity)
__global__ void probe(const double* a, const double* b, double* d) {
  int l = threadIdx.x;
  double r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
  d[l] = r;
}
int main() {
  double ha[64], hb[64], hd[64];
  double *da, *db, *dd;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  // experiment 1: A lane l gets value l (encodes lane), B = all ones in lane pattern to find which lanes pair
  for (int t = 0; t < 3; t++) {
    for (int l = 0; l < 64; l++) { ha[l] = 0; hb[l] = 0; }
    if (t == 0) { for (int l = 0; l < 64; l++) { ha[l] = 1.0; hb[l] = (double)(1 << (l % 16)) ; } }   // D = sum over k of B -> which B lanes feed each D lane
    if (t == 1) { for (int l = 0; l < 64; l++) { hb[l] = 1.0; ha[l] = (double)(1 << (l % 16)) ; } }
    if (t == 2) { for (int l = 0; l < 64; l++) { ha[l] = l; hb[l] = 1000.0 * l; } }
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    printf("exp %d:\n", t);
    for (int l = 0; l < 64; l++) { printf("%g ", hd[l]); if (l % 16 == 15) printf("\n"); }
  }
  return 0;
}
