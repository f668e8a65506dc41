// Test helper (tests/test_gpu_unity_plugin.py): stands in for the GRAPHICS API of the texture interop.  It allocates device memory with HIP's
// virtual-memory API, exports it as a POSIX fd (a dma-buf, like Vulkan's vkGetMemoryFdKHR on Linux) and keeps its own mapping of it -- the
// "render texture" the host would sample.  Built by the test with g++ against libamdhip64 (host code only).
#include <hip/hip_runtime_api.h>
#include <cstring>
extern "C" {
int producer_create(size_t bytes, int* fd_out, void** view_out, size_t* bytes_out)
{
    hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) return 1;
    bytes = (bytes + gran - 1) / gran * gran;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, bytes, &prop, 0) != hipSuccess) return 2;
    if (hipMemExportToShareableHandle(fd_out, h, hipMemHandleTypePosixFileDescriptor, 0) != hipSuccess) return 3;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, bytes, 0, nullptr, 0) != hipSuccess || hipMemMap(va, bytes, 0, h, 0) != hipSuccess) return 4;
    hipMemAccessDesc acc; memset(&acc, 0, sizeof acc); acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, bytes, &acc, 1) != hipSuccess) return 5;
    if (hipMemset(va, 0, bytes) != hipSuccess) return 6;
    *view_out = va; *bytes_out = bytes;
    return 0;
}
int producer_read(const void* view, size_t offset, void* host, size_t bytes)
{
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpy(host, (const char*)view + offset, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
}
