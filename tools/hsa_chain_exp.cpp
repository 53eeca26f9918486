// hsa_chain_exp.cpp -- EXPERIMENT (not product code): dispatch the library's step kernel through an HSA queue of our own,
// with explicit acquire / release fence scopes on the AQL packets, to measure what the agent-scope release + acquire that HIP
// attaches to every launch costs a chain of dependent step launches.
//   build: g++ -O2 -fPIC -shared tools/hsa_chain_exp.cpp -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -o tools/libqs_hsa_exp.so
//   use:   tools/hsa_chain_exp.py
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

struct Ctx {
    hsa_agent_t gpu{};
    bool have_gpu = false;
    hsa_amd_memory_pool_t kernarg_pool{};
    bool have_pool = false;
    hsa_queue_t *queue = nullptr;
    hsa_executable_t exe{};
    hsa_code_object_reader_t reader{};
    uint64_t kernel_object = 0;
    uint32_t kernarg_size = 0, group_size = 0, private_size = 0;
    void *kernargs = nullptr;
    size_t kernarg_stride = 0, kernarg_slots = 0;
    hsa_signal_t done{};
    int gpu_index_wanted = 0, gpu_seen = 0;
};

Ctx g;
char g_err[256] = "";

#define CK(x)                                                               \
    do {                                                                    \
        hsa_status_t s_ = (x);                                              \
        if (s_ != HSA_STATUS_SUCCESS) {                                     \
            const char *m_ = nullptr;                                       \
            hsa_status_string(s_, &m_);                                     \
            snprintf(g_err, sizeof g_err, "%s -> %s", #x, m_ ? m_ : "?");  \
            return -1;                                                      \
        }                                                                   \
    } while (0)

hsa_status_t agent_cb(hsa_agent_t a, void *)
{
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_GPU) {
        if (g.gpu_seen++ == g.gpu_index_wanted) { g.gpu = a; g.have_gpu = true; }
    }
    return HSA_STATUS_SUCCESS;
}

hsa_status_t pool_cb(hsa_amd_memory_pool_t p, void *)
{
    hsa_amd_segment_t seg;
    if (hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    bool alloc = false;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    // device-local coarse-grained memory: what HIP itself uses for kernargs on this platform
    if (alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g.have_pool) { g.kernarg_pool = p; g.have_pool = true; }
    return HSA_STATUS_SUCCESS;
}

}  // namespace

extern "C" {

const char *qsx_error() { return g_err; }

// load the code object, find the kernel, create the queue and a ring of kernarg blocks
int qsx_open(const char *hsaco_path, const char *kernel_symbol, int gpu_index, uint32_t slots)
{
    CK(hsa_init());
    g.gpu_index_wanted = gpu_index;
    g.gpu_seen = 0;
    CK(hsa_iterate_agents(agent_cb, nullptr));
    if (!g.have_gpu) { snprintf(g_err, sizeof g_err, "no GPU agent"); return -1; }
    CK(hsa_amd_agent_iterate_memory_pools(g.gpu, pool_cb, nullptr));
    if (!g.have_pool) { snprintf(g_err, sizeof g_err, "no device memory pool"); return -1; }
    hsa_file_t fd = open(hsaco_path, O_RDONLY);
    if (fd < 0) { snprintf(g_err, sizeof g_err, "cannot open %s", hsaco_path); return -1; }
    CK(hsa_code_object_reader_create_from_file(fd, &g.reader));
    CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &g.exe));
    CK(hsa_executable_load_agent_code_object(g.exe, g.gpu, g.reader, nullptr, nullptr));
    CK(hsa_executable_freeze(g.exe, nullptr));
    hsa_executable_symbol_t sym;
    CK(hsa_executable_get_symbol_by_name(g.exe, kernel_symbol, &g.gpu, &sym));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &g.kernel_object));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &g.kernarg_size));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &g.group_size));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &g.private_size));
    CK(hsa_queue_create(g.gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &g.queue));
    g.kernarg_stride = (g.kernarg_size + 255) & ~size_t(255);
    g.kernarg_slots = slots;
    CK(hsa_amd_memory_pool_allocate(g.kernarg_pool, g.kernarg_stride * slots, 0, &g.kernargs));
    CK(hsa_signal_create(1, 0, nullptr, &g.done));
    return 0;
}

int qsx_info(uint32_t *kernarg_size, uint32_t *group_size, uint32_t *private_size)
{
    *kernarg_size = g.kernarg_size; *group_size = g.group_size; *private_size = g.private_size;
    return 0;
}

// host-side write of kernarg slot i (device memory: through the HSA copy; done once before the timed run)
int qsx_set_kernarg(uint32_t slot, const void *bytes, uint32_t size)
{
    if (slot >= g.kernarg_slots || size > g.kernarg_size) { snprintf(g_err, sizeof g_err, "bad kernarg slot / size"); return -1; }
    std::vector<char> tmp(g.kernarg_stride, 0);
    memcpy(tmp.data(), bytes, size);
    CK(hsa_memory_copy((char *)g.kernargs + slot * g.kernarg_stride, tmp.data(), g.kernarg_stride));
    return 0;
}

// K dispatches of the kernel, packet k using kernarg slot k % slots; every packet has the barrier bit (a dependent chain)
// and the given fence scopes (0 none, 1 agent, 2 system); the LAST packet releases at system scope and carries the signal.
// Returns the wall time in microseconds between ringing the first doorbell and the completion of the last packet.
// a repeating pattern of packets: packet j uses kernarg slot pat_slot[j % pat_len] and grid pat_grid[j % pat_len]
static uint32_t g_pat_len = 0, g_pat_slot[64], g_pat_grid[64];
int qsx_set_pattern(uint32_t len, const uint32_t *slots, const uint32_t *grids)
{
    if (len > 64) return -1;
    g_pat_len = len;
    for (uint32_t i = 0; i < len; ++i) { g_pat_slot[i] = slots[i]; g_pat_grid[i] = grids[i]; }
    return 0;
}

int qsx_run_chain(uint32_t K, uint32_t grid_x, uint32_t block_x, int acquire_scope, int release_scope, double *elapsed_us)
{
    if (!g.queue) { snprintf(g_err, sizeof g_err, "not open"); return -1; }
    hsa_signal_store_relaxed(g.done, 1);
    const uint32_t mask = g.queue->size - 1;
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t k = 0; k < K; ++k) {
        uint64_t idx = hsa_queue_add_write_index_relaxed(g.queue, 1);
        while (idx - hsa_queue_load_read_index_scacquire(g.queue) >= g.queue->size) {}
        hsa_kernel_dispatch_packet_t *p = (hsa_kernel_dispatch_packet_t *)g.queue->base_address + (idx & mask);
        const bool last = k + 1 == K;
        p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        p->workgroup_size_x = (uint16_t)block_x; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
        p->grid_size_x = g_pat_len ? g_pat_grid[k % g_pat_len] : grid_x; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = g.private_size;
        p->group_segment_size = g.group_size;
        p->kernel_object = g.kernel_object;
        p->kernarg_address = (char *)g.kernargs + ((g_pat_len ? g_pat_slot[k % g_pat_len] : k) % g.kernarg_slots) * g.kernarg_stride;
        p->reserved2 = 0;
        p->completion_signal = last ? g.done : hsa_signal_t{0};
        const int acq = k == 0 ? 2 : acquire_scope;
        const int rel = last ? 2 : release_scope;
        uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                          (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        __atomic_store_n(&p->header, header, __ATOMIC_RELEASE);
        hsa_signal_store_screlease(g.queue->doorbell_signal, idx);
    }
    while (hsa_signal_wait_scacquire(g.done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) {}
    auto t1 = std::chrono::steady_clock::now();
    *elapsed_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
    return 0;
}

// the same chain split over Q queues: queue q owns kernarg slots [q*per, (q+1)*per) (prepared by the caller with the tile range
// of group q) and gets one packet per step; every queue ends with a system-scope release + signal
int qsx_run_chain_multi(uint32_t K, uint32_t Q, uint32_t per, uint32_t grid_x, uint32_t block_x, int acquire_scope, int release_scope,
                        double *elapsed_us)
{
    static hsa_queue_t *qs[8] = {nullptr};
    static hsa_signal_t sig[8];
    if (Q > 8) return -1;
    for (uint32_t q = 0; q < Q; ++q)
        if (!qs[q]) {
            CK(hsa_queue_create(g.gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &qs[q]));
            CK(hsa_signal_create(1, 0, nullptr, &sig[q]));
        }
    for (uint32_t q = 0; q < Q; ++q) hsa_signal_store_relaxed(sig[q], 1);
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t k = 0; k < K; ++k)
        for (uint32_t q = 0; q < Q; ++q) {
            hsa_queue_t *Qu = qs[q];
            uint64_t idx = hsa_queue_add_write_index_relaxed(Qu, 1);
            while (idx - hsa_queue_load_read_index_scacquire(Qu) >= Qu->size) {}
            hsa_kernel_dispatch_packet_t *p = (hsa_kernel_dispatch_packet_t *)Qu->base_address + (idx & (Qu->size - 1));
            const bool last = k + 1 == K;
            p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            p->workgroup_size_x = (uint16_t)block_x; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
            p->grid_size_x = grid_x; p->grid_size_y = 1; p->grid_size_z = 1;
            p->private_segment_size = g.private_size;
            p->group_segment_size = g.group_size;
            p->kernel_object = g.kernel_object;
            p->kernarg_address = (char *)g.kernargs + (q * per + k % per) * g.kernarg_stride;
            p->reserved2 = 0;
            p->completion_signal = last ? sig[q] : hsa_signal_t{0};
            const int acq = k == 0 ? 2 : acquire_scope;
            const int rel = last ? 2 : release_scope;
            uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                              (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
            __atomic_store_n(&p->header, header, __ATOMIC_RELEASE);
            hsa_signal_store_screlease(Qu->doorbell_signal, idx);
        }
    for (uint32_t q = 0; q < Q; ++q)
        while (hsa_signal_wait_scacquire(sig[q], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) {}
    auto t1 = std::chrono::steady_clock::now();
    *elapsed_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
    return 0;
}

int qsx_close()
{
    if (g.queue) hsa_queue_destroy(g.queue);
    if (g.kernargs) hsa_amd_memory_pool_free(g.kernargs);
    g.queue = nullptr; g.kernargs = nullptr;
    return 0;
}

}  // extern "C"
