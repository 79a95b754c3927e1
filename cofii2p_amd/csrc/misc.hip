// ABI identification of libcofi_hip.so.
#include "common.h"

extern "C" int cofi_abi_version(void) { return COFI_ABI_VERSION; }
extern "C" const char *cofi_target_arch(void) { return "gfx950"; }

// Batched device-to-device copy: ONE launch moves a whole list of tensors (the per-frame inputs into the static buffers a
// captured hipGraph reads).  20+ separate copies cost ~3 us of stream time each; the descriptors travel as one small H2D copy.
namespace {
struct CopyDesc {
    const void *src;
    void *dst;
    unsigned long long bytes;
};

__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyDesc *descs) {
    const CopyDesc d = descs[blockIdx.y];
    const size_t n16 = (((uintptr_t)d.src | (uintptr_t)d.dst) & 15) ? 0 : d.bytes / 16;
    const uint4 *s4 = reinterpret_cast<const uint4 *>(d.src);
    uint4 *d4 = reinterpret_cast<uint4 *>(d.dst);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // four independent 16-byte loads in flight per lane: the 10 MB neighbour table of a frame is latency-bound otherwise
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 v0 = s4[i], v1 = s4[i + stride], v2 = s4[i + 2 * stride], v3 = s4[i + 3 * stride];
        d4[i] = v0; d4[i + stride] = v1; d4[i + 2 * stride] = v2; d4[i + 3 * stride] = v3;
    }
    for (; i < n16; i += stride) d4[i] = s4[i];
    const unsigned char *sb = reinterpret_cast<const unsigned char *>(d.src);
    unsigned char *db = reinterpret_cast<unsigned char *>(d.dst);
    for (size_t b = n16 * 16 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < d.bytes; b += stride) db[b] = sb[b];
}
}  // namespace

extern "C" int cofi_multi_copy(const void *descs_dev, int n, int blocks_per_copy, cofi_stream_t stream) {
    if (!descs_dev || n < 0 || blocks_per_copy <= 0) return COFI_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(multi_copy_kernel, dim3(blocks_per_copy, n), dim3(256), 0, cofi_s(stream), (const CopyDesc *)descs_dev);
    return cofi_launch_status();
}
