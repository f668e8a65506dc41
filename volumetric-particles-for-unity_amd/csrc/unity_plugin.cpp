// unity_plugin.cpp -- the Unity low-level native-plugin hookup of the component that replaces VolumetricParticleRenderer
// (SURVEY 8(f) row 3).  The reference issues its fill / ray-march as Graphics.Blit / DrawMeshNow from OnPostRender on the main thread
// (VPR.cs:181-220); a native plugin does its GPU work on Unity's RENDER thread instead: C# fills a frame description, calls
// GL.IssuePluginEvent(vp_unity_render_event_func(), slot), and Unity invokes the callback on the render thread, where the frame runs
//     [vp_set_frame] -> [vp_bin -> vp_fill] -> vp_raymarch  (bin + fill only when the description says so: updateInterval, VPR.cs:186)
// and the premultiplied RGBA lands in the output the host registered for the slot: a HIP device pointer (vp_raymarch_device writes it; how
// the host obtains one for its render texture -- e.g. hipImportExternalMemory over the texture's exported allocation -- is its business)
// and / or a host buffer.  The status of the last event is polled with vp_unity_last_status (Unity's callback returns void).
// The entry points follow the shape of Unity's public plugin API (IUnityInterface.h: UnityPluginLoad / UnityPluginUnload exported by
// name, a `void (*)(int eventId)` rendering event); Unity's own headers are not needed and not used: IUnityInterfaces* is kept as an
// opaque pointer.  Nothing here is Unity-specific beyond that calling convention, so the tests drive it from a second host thread.
#include <condition_variable>
#include <cstring>
#include <mutex>

#include "vpfx_internal.h"

namespace {

struct Slot {
    vp_unity_frame frame{};
    bool have_frame = false;
    void* d_out = nullptr;
    float* h_out = nullptr;
    int status = VP_ERR_STATE;
    unsigned long long events = 0;
    bool running = false;         // an event of this slot is executing (vp_unity_clear_slot waits for it)
    hipExternalMemory_t ext = nullptr;   // vp_unity_register_output_fd: the imported allocation d_out points into
    int ext_device = 0;
    void* ext_ptr = nullptr;             // the mapping of that import (d_out while the import is the slot's device output)
};
std::mutex g_m;
std::condition_variable g_cv;
Slot g_slots[VP_UNITY_MAX_SLOTS];
void* g_unity_interfaces = nullptr;
bool g_loaded = false;

// the mapping of an imported allocation dies with the import: drop both (the slot's d_out pointed into it)
void release_external(Slot& s)
{
    if (!s.ext) return;
    (void)hipSetDevice(s.ext_device);
    (void)hipDeviceSynchronize();                     // nothing of ours may still be writing the shared memory
    (void)hipDestroyExternalMemory(s.ext);
    if (s.d_out == s.ext_ptr) s.d_out = nullptr;      // only the output that pointed INTO the import dies with it (a device buffer
    s.ext = nullptr; s.ext_ptr = nullptr;             // registered later must survive the release of an older import)
}

void on_render_event(int slot)
{
    if (slot < 0 || slot >= VP_UNITY_MAX_SLOTS) return;
    vp_unity_frame f;
    void* d_out;
    float* h_out;
    {
        std::lock_guard<std::mutex> lk(g_m);
        if (!g_slots[slot].have_frame) { g_slots[slot].status = VP_ERR_STATE; ++g_slots[slot].events; return; }     // (also: an event that arrives after vp_unity_clear_slot)
        f = g_slots[slot].frame; d_out = g_slots[slot].d_out; h_out = g_slots[slot].h_out;
        g_slots[slot].running = true;
    }
    int rc = VP_OK;
    vp_ctx* c = f.ctx;
    if (!c) rc = VP_ERR_BAD_ARG;
    if (!rc && (f.flags & VP_UNITY_SET_FRAME)) rc = vp_set_frame(c, f.light_to_world, f.grid_center);               // VPR.cs:188-195
    if (!rc && (f.flags & VP_UNITY_BIN_AND_FILL)) {                                                                   // VPR.cs:186-201
        rc = vp_bin(c, f.particles, f.particle_count, &f.layout, f.psys_local_to_world);
        if (!rc) rc = vp_fill(c, &f.fill);
    }
    if (!rc) {                                                                                                        // VPR.cs:207
        if (d_out) rc = vp_raymarch_device(c, &f.camera, &f.raymarch, d_out);
        if (!rc && h_out) rc = d_out ? vp_read_last_image(c, d_out, h_out) : vp_raymarch(c, &f.camera, &f.raymarch, h_out);
        if (!rc && !d_out && !h_out) rc = VP_ERR_STATE;                  // nowhere to put the frame: vp_unity_register_output first
    }
    std::lock_guard<std::mutex> lk(g_m);
    g_slots[slot].status = rc;
    ++g_slots[slot].events;
    g_slots[slot].running = false;
    g_cv.notify_all();
}

}  // namespace

// read an image the ray-march left on the device back to the host (the host-buffer half of a slot that has both outputs)
int vp_read_last_image(vp_ctx* c, const void* d_img, float* h_out)
{
    if (c->multi) c = multi_owner_of_slice(c, -1);
    if (!c) return VP_ERR_STATE;
    VP_HIP(hipSetDevice(c->device));
    VP_HIP(hipMemcpyAsync(h_out, d_img, (size_t)c->cfg.width * c->cfg.height * 4 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    return api_stream_sync(c);
}

VP_EXPORT void UnityPluginLoad(void* unity_interfaces)
{
    std::lock_guard<std::mutex> lk(g_m);
    g_unity_interfaces = unity_interfaces;
    g_loaded = true;
}

VP_EXPORT void UnityPluginUnload(void)
{
    std::unique_lock<std::mutex> lk(g_m);
    g_cv.wait(lk, [&] { for (const Slot& s : g_slots) if (s.running) return false; return true; });      // an event that is executing keeps its import
    for (Slot& s : g_slots) { release_external(s); s = Slot{}; }
    g_unity_interfaces = nullptr;
    g_loaded = false;
}

VP_EXPORT vp_unity_render_event vp_unity_render_event_func(void) { return on_render_event; }

VP_EXPORT int vp_unity_set_frame_desc(int32_t slot, const vp_unity_frame* frame)
{
    if (slot < 0 || slot >= VP_UNITY_MAX_SLOTS || !frame || !frame->ctx) return VP_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_m);
    g_slots[slot].frame = *frame;          // copied: the caller's struct need not outlive the call (the arrays it points to must live until the event has run)
    g_slots[slot].have_frame = true;
    return VP_OK;
}

VP_EXPORT int vp_unity_register_output(int32_t slot, void* d_rgba_out, float* h_rgba_out)
{
    if (slot < 0 || slot >= VP_UNITY_MAX_SLOTS) return VP_ERR_BAD_ARG;
    std::unique_lock<std::mutex> lk(g_m);
    g_cv.wait(lk, [&] { return !g_slots[slot].running; });      // an event of the slot that is executing right now keeps its outputs
    release_external(g_slots[slot]);                             // a previous vp_unity_register_output_fd import is replaced, not leaked
    g_slots[slot].d_out = d_rgba_out;
    g_slots[slot].h_out = h_rgba_out;
    return VP_OK;
}

// Texture interop, the native half (VPR.cs:204-210 blits particlesRT on the GPU -- here the frame crossed PCIe).  The graphics API
// exports the memory behind the texture's linear RGBA32F buffer as a POSIX fd (Vulkan: VkExportMemoryAllocateInfo + vkGetMemoryFdKHR, opaque fd or
// dma-buf); this imports it through HIP's external-memory API on the context's (display) device, maps `W H 16` bytes at `offset`, and registers the
// mapping as the slot's DEVICE output: the ray-march then writes the shared memory itself, no copy.  The fd belongs to HIP after a successful import
// (do not close it).  vp_unity_clear_slot / UnityPluginUnload drop the import.  Ordering against the graphics queue is the host's business (a
// timeline semaphore imported with hipImportExternalSemaphore, waited / signalled on the stream given to vp_set_stream: INTEGRATION.md).
// Tested with a dma-buf fd exported by HIP's own virtual-memory API standing in for the graphics API (tests/test_gpu_unity_plugin.py).
VP_EXPORT int vp_unity_register_output_fd(int32_t slot, vp_ctx* ctx, int32_t fd, uint64_t bytes, uint64_t offset)
{
    if (slot < 0 || slot >= VP_UNITY_MAX_SLOTS || !ctx) return VP_ERR_BAD_ARG;
    vp_ctx* c = ctx->multi ? multi_owner_of_slice(ctx, -1) : ctx;
    if (!c) return vp_fail(ctx, VP_ERR_STATE, "vp_unity_register_output_fd: the display rank is not on this process");
    const uint64_t need = (uint64_t)c->cfg.width * c->cfg.height * 4 * sizeof(float);
    if (fd < 0 || offset > bytes || bytes - offset < need)
        return vp_fail(ctx, VP_ERR_BAD_ARG, "vp_unity_register_output_fd: %llu bytes at offset %llu of a %llu-byte allocation (particlesRT needs %llu)",
                       (unsigned long long)(bytes - (offset > bytes ? bytes : offset)), (unsigned long long)offset, (unsigned long long)bytes, (unsigned long long)need);
    if (hipSetDevice(c->device) != hipSuccess) return vp_fail(ctx, VP_ERR_HIP, "hipSetDevice failed");
    hipExternalMemoryHandleDesc d;
    memset(&d, 0, sizeof d);
    d.type = hipExternalMemoryHandleTypeOpaqueFd; d.handle.fd = fd; d.size = bytes;
    hipExternalMemory_t ext = nullptr;
    hipError_t e = hipImportExternalMemory(&ext, &d);
    if (e != hipSuccess) { (void)hipGetLastError(); return vp_fail(ctx, VP_ERR_HIP, "hipImportExternalMemory failed: %s", hipGetErrorString(e)); }
    hipExternalMemoryBufferDesc b;
    memset(&b, 0, sizeof b);
    b.offset = offset; b.size = need;
    void* dptr = nullptr;
    e = hipExternalMemoryGetMappedBuffer(&dptr, ext, &b);
    if (e != hipSuccess) { (void)hipGetLastError(); (void)hipDestroyExternalMemory(ext); return vp_fail(ctx, VP_ERR_HIP, "hipExternalMemoryGetMappedBuffer failed: %s", hipGetErrorString(e)); }
    std::unique_lock<std::mutex> lk(g_m);
    g_cv.wait(lk, [&] { return !g_slots[slot].running; });
    release_external(g_slots[slot]);
    g_slots[slot].ext = ext; g_slots[slot].ext_device = c->device; g_slots[slot].ext_ptr = dptr;
    g_slots[slot].d_out = dptr;
    return VP_OK;
}

VP_EXPORT int vp_unity_last_status(int32_t slot, uint64_t* events_run)
{
    if (slot < 0 || slot >= VP_UNITY_MAX_SLOTS) return VP_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_m);
    if (events_run) *events_run = g_slots[slot].events;
    return g_slots[slot].status;
}

// Detach a slot from its context and buffers: waits for an event of the slot that is executing right now, then forgets the frame
// description and the outputs -- an event Unity delivers later finds nothing and is a no-op (status VP_ERR_STATE).  The host calls this
// BEFORE it frees the arrays the description points to and before vp_destroy: the render thread may run an issued event at any later time.
VP_EXPORT int vp_unity_clear_slot(int32_t slot)
{
    if (slot < 0 || slot >= VP_UNITY_MAX_SLOTS) return VP_ERR_BAD_ARG;
    std::unique_lock<std::mutex> lk(g_m);
    g_cv.wait(lk, [&] { return !g_slots[slot].running; });
    const unsigned long long ev = g_slots[slot].events;
    release_external(g_slots[slot]);
    g_slots[slot] = Slot{};
    g_slots[slot].events = ev;                 // the counter keeps counting: hosts compare it with the number of events they issued
    return VP_OK;
}

