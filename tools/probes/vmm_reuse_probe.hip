// vmm_reuse_probe.hip — does a virtual-memory mapping made at an address range that was reserved, mapped, unmapped and freed
// just before see its own memory?  (No code of libchalkydri_hip involved: this is what CK_POISON=3's guard-page allocator and
// ck_selftest_fp64's three back-to-back allocations do, reduced to the runtime calls.)
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/probes/vmm_reuse_probe.hip -o tools/probes/vmm_reuse_probe
//   tools/probes/vmm_reuse_probe [rounds] [non-blocking stream 0/1] [keep the address ranges reserved 0/1]
// Measured on MI355X / ROCm 7.2 (gpurun box): with the ranges freed, every round after the first gets the previous round's
// addresses back and the kernel's output reads as zeros (fresh pages) or as the 0xA5 fill (the copy engine's view of the new
// pages): the kernel wrote through translations of the OLD, released pages.  With the ranges kept reserved (third argument 1)
// no address comes back and every round is clean.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#define OK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void k_axpy(const double *a, const double *b, int n, double *out) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = a[i] * b[i] + a[i]; }
struct Map { void *va; size_t va_bytes, map_bytes; hipMemGenericAllocationHandle_t mem; double *p; };
static int make(Map &m, size_t bytes, size_t gran, const hipMemAllocationProp &prop) {
    m.map_bytes = (bytes + gran - 1) / gran * gran; m.va_bytes = m.map_bytes + gran; // one granule stays unmapped behind the buffer
    OK(hipMemAddressReserve(&m.va, m.va_bytes, gran, nullptr, 0));
    OK(hipMemCreate(&m.mem, m.map_bytes, &prop, 0));
    OK(hipMemMap(m.va, m.map_bytes, 0, m.mem, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    OK(hipMemSetAccess(m.va, m.map_bytes, &acc, 1));
    m.p = reinterpret_cast<double *>(static_cast<char *>(m.va) + (m.map_bytes - bytes)); // the buffer ENDS at the end of the mapping
    return 0;
}
static bool g_keep_va = false;
static int drop(Map &m) { OK(hipDeviceSynchronize()); OK(hipMemUnmap(m.va, m.map_bytes)); OK(hipMemRelease(m.mem)); if (!g_keep_va) OK(hipMemAddressFree(m.va, m.va_bytes)); return 0; }
int main(int argc, char **argv) {
    const int n = 500000, rounds = argc > 1 ? atoi(argv[1]) : 6;
    const bool nonblocking = argc > 2 ? atoi(argv[2]) != 0 : true;
    g_keep_va = argc > 3 ? atoi(argv[3]) != 0 : false;
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; OK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    hipStream_t st; OK(hipStreamCreateWithFlags(&st, nonblocking ? hipStreamNonBlocking : hipStreamDefault));
    std::vector<double> a(n), b(n), out(n);
    const size_t bytes = sizeof(double) * n;
    int bad_rounds = 0;
    for (int r = 0; r < rounds; r++) {
        for (int i = 0; i < n; i++) { a[i] = 1.0 + i * 1e-3 + r; b[i] = 2.0 - i * 1e-4; }
        Map ma, mb, mo;
        if (make(ma, bytes, gran, prop) || make(mo, bytes, gran, prop) || make(mb, bytes, gran, prop)) return 2;
        OK(hipMemset(ma.p, 0xA5, bytes)); OK(hipMemset(mo.p, 0xA5, bytes)); OK(hipMemset(mb.p, 0xA5, bytes)); OK(hipDeviceSynchronize()); // (CK_POISON's fill)
        OK(hipMemcpy(ma.p, a.data(), bytes, hipMemcpyHostToDevice));
        OK(hipMemcpy(mb.p, b.data(), bytes, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_axpy, dim3((n + 255) / 256), dim3(256), 0, st, ma.p, mb.p, n, mo.p);
        OK(hipStreamSynchronize(st));
        OK(hipMemcpy(out.data(), mo.p, bytes, hipMemcpyDeviceToHost));
        long bad = 0, zeros = 0, poison = 0;
        for (int i = 0; i < n; i++) {
            const double want = a[i] * b[i] + a[i];
            if (out[i] != want) { bad++; zeros += out[i] == 0.0; unsigned long long u; memcpy(&u, &out[i], 8); poison += u == 0xA5A5A5A5A5A5A5A5ull; }
        }
        printf("round %d: va a=%p out=%p b=%p  wrong %ld (zeros %ld, poison pattern %ld)\n", r, (void *)ma.p, (void *)mo.p, (void *)mb.p, bad, zeros, poison);
        bad_rounds += bad != 0;
        if (drop(ma) || drop(mb) || drop(mo)) return 2;
    }
    printf("%s: %d of %d rounds returned wrong data\n", bad_rounds ? "REPRODUCED" : "clean", bad_rounds, rounds);
    return bad_rounds ? 1 : 0;
}
