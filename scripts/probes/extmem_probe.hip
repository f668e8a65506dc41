// extmem_probe.hip -- can HIP import, through the EXTERNAL-MEMORY API (what a graphics API's exported allocation goes through), a dma-buf fd
// that HIP's own virtual-memory API exported?  (The box has no Vulkan / D3D: this stands in for "Unity's render texture memory exported as an fd".)
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/extmem_probe scripts/probes/extmem_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    const size_t bytes = 64u << 20;
    hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, bytes, &prop, 0));
    int fd = -1; CK(hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0));
    printf("exported fd %d (granularity %zu)\n", fd, gran);
    // the producer's own view of the memory
    void* va = nullptr; CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0)); CK(hipMemMap(va, bytes, 0, h, 0));
    hipMemAccessDesc acc; memset(&acc, 0, sizeof acc); acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, bytes, &acc, 1));
    CK(hipMemset(va, 0, bytes));
    // the consumer: external-memory import of that fd
    hipExternalMemoryHandleDesc d; memset(&d, 0, sizeof d);
    d.type = hipExternalMemoryHandleTypeOpaqueFd; d.handle.fd = fd; d.size = bytes;
    hipExternalMemory_t ext; CK(hipImportExternalMemory(&ext, &d));
    hipExternalMemoryBufferDesc b; memset(&b, 0, sizeof b); b.offset = 0; b.size = bytes;
    void* dptr = nullptr; CK(hipExternalMemoryGetMappedBuffer(&dptr, ext, &b));
    CK(hipMemset(dptr, 0x5a, 4096));
    CK(hipDeviceSynchronize());
    std::vector<unsigned char> host(4096);
    CK(hipMemcpy(host.data(), va, 4096, hipMemcpyDeviceToHost));
    printf("imported %p; producer view reads 0x%02x 0x%02x -> %s\n", dptr, host[0], host[4095], host[0] == 0x5a && host[4095] == 0x5a ? "SAME MEMORY" : "different memory");
    CK(hipDestroyExternalMemory(ext));
    return 0;
}
