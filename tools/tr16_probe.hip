// tr16_probe.hip - pins the lane / element mapping of ds_read_b64_tr_b16 (gfx950) on hardware, for the NHWC-fed weight-gradient
// kernel (csrc/wgrad.hip): LDS holds halves whose value is their own index; every lane passes an address, the probe returns the
// four halves each lane received.  Build: hipcc --offload-arch=gfx950 -shared -fPIC -o tr16_probe.so tr16_probe.hip
#include <hip/hip_runtime.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void tr16_probe_kernel(const int* __restrict__ byte_addr, short* __restrict__ out, int nhalves) {
    extern __shared__ short lds[];
    for (int i = threadIdx.x; i < nhalves; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)((__attribute__((address_space(3))) char*)lds + byte_addr[l]));
    *(s4*)(out + l * 4) = v;
}
extern "C" int tr16_probe(const int* byte_addr, short* out, int nhalves, void* stream) {
    hipLaunchKernelGGL(tr16_probe_kernel, dim3(1), dim3(64), (size_t)nhalves * 2, (hipStream_t)stream, byte_addr, out, nhalves);
    return (int)hipGetLastError();
}
