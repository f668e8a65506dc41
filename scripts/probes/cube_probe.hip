// Probe of the gfx950 cube-map VALU instructions (v_cubeid/sc/tc/ma_f32): exact semantics incl. ties and signs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* in, float* out, int n) {
    int i = blockIdx.x * 64 + threadIdx.x; if (i >= n) return;
    float x = in[3*i], y = in[3*i+1], z = in[3*i+2];
    out[4*i]   = __builtin_amdgcn_cubeid(x, y, z);
    out[4*i+1] = __builtin_amdgcn_cubesc(x, y, z);
    out[4*i+2] = __builtin_amdgcn_cubetc(x, y, z);
    out[4*i+3] = __builtin_amdgcn_cubema(x, y, z);
}
int main() {
    std::vector<float> v = {
        1,0.5f,0.25f,  -1,0.5f,0.25f,  0.5f,1,0.25f,  0.5f,-1,0.25f,  0.5f,0.25f,1,  0.5f,0.25f,-1,
        1,1,0.5f,  1,0.5f,1,  0.5f,1,1,  1,1,1,  -1,-1,-1,  -1,1,1, 1,-1,1, 1,1,-1, -1,-1,0.5f, -1,0.5f,-1, 0.5f,-1,-1,
        0,0,0,  -0.f,0,0,  0,0,1e-30f, 3,-2,-1, -0.3f,0.2f,-0.25f };
    int n = v.size() / 3;
    float *di, *dout; hipMalloc(&di, v.size()*4); hipMalloc(&dout, n*16);
    hipMemcpy(di, v.data(), v.size()*4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n+63)/64), dim3(64), 0, 0, di, dout, n);
    std::vector<float> o(n*4); hipMemcpy(o.data(), dout, n*16, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("(% .3g,% .3g,% .3g) -> id %g sc % g tc % g ma % g\n", v[3*i], v[3*i+1], v[3*i+2], o[4*i], o[4*i+1], o[4*i+2], o[4*i+3]);
    return 0;
}
