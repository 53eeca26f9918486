// private_queue.hpp -- the handle's private AQL queues (qs_set_queue_mode): step launches without the end-of-kernel release
// A fragment of quadsim_hip.hip (ONE translation unit: the kernels' mangled names, which the private-queue code resolves
// in the code object, live in that unit's anonymous namespace); included there at a fixed position, nowhere else.
#pragma once

// ---- private AQL queue for the step launches (qs_set_queue_mode) -------------------------------------------------------
// Every kernel HIP launches ends with an agent-scope release: the eight XCD L2s are not coherent with each other, so their
// dirty lines are written back before the next packet may start.  For a chain of dependent step launches that write-back is
// 1.6 of 6.5 us per step at 65 536 envs (profiles/r02/ab_experiments.txt, section E) -- and it is not needed: tile b is
// stepped by workgroup b of every launch, workgroup b always lands on the same XCD, so the tile's state can stay dirty in
// that XCD's L2 from one step to the next.  HIP has no launch without the fence; an AQL packet written by hand has:
//     header = KERNEL_DISPATCH | BARRIER (ordered behind the previous packet) | ACQUIRE agent (fresh kernargs / actions;
//              0.16 us, does not touch dirty lines) | RELEASE none.
// The handle therefore owns an HSA queue, loads its own copy of the library's code object into it, and writes one packet per
// qs_step.  Everything else stays on HIP: any other entry point first DRAINS the queue with a release packet (host wait).
//
// Ordering against the CALLER's work (the handle's stream S), round 3 -- QS_ORDER_STREAM, the default where the device has
// stream memory operations: a step behaves like a launch on S although it runs elsewhere.  Per submission (one qs_step, or the
// T steps of qs_rollout_stepwise):
//     S:      hipStreamWriteValue64(fwd, n)              -- executes when everything enqueued on S so far has finished
//     queue:  barrier-value packet (fwd >= n) | step packet(s), the last one with a completion signal `rev`
//     S:      hipStreamWaitValue64(rev == V0 - n)        -- whatever is enqueued on S afterwards runs after the step(s)
// and that last packet carries an agent-scope RELEASE, so that the outputs of all T steps are in memory -- not dirty in one XCD's
// L2 -- when `rev` fires (the state lines are written back with them and stay valid in their L2).  No host synchronisation
// anywhere.  A per-step loop (T = 1) thereby pays what a HIP launch pays -- per-step consumable outputs ARE the write-back:
// write-through output stores instead of the release were measured no faster (the kernel then ends when memory, not the L2,
// acknowledges its stores) -- plus the hand-shake; a T-step roll-out pays both once.  QS_ORDER_HOST is round 2's contract:
// inputs complete at the call, outputs valid after qs_sync (no hand-shake packets).
struct QsChainLane {                  // one private queue and the contiguous tile range it steps
    hsa_queue_t *queue = nullptr;
    char *kernargs = nullptr;
    std::vector<uint64_t> slot_qidx;  // queue index of the packet that last used each kernarg slot
    uint64_t issued = 0;              // step packets written so far
    int64_t tile0 = 0, tile_end = 0;
    hsa_signal_t done{};
    // QS_ORDER_STREAM: completion signal of this lane's submissions.  Allocated by HIP as "signal memory" (the only memory
    // hipStreamWaitValue64 accepts); HIP hands out the address of the signal's VALUE, the handle is the amd_signal_t around it.
    void *rev_ptr = nullptr;
    hsa_signal_t rev{};
    int64_t rev_value = 0;            // value of `rev` once every submission so far has completed (counts DOWN: AQL decrements)
};

struct QsChain {
    hsa_agent_t gpu{}, cpu{};
    hsa_amd_memory_pool_t kernarg_pool{};
    std::vector<QsChainLane> lanes;
    int requested = 0;                // the queue count qs_set_queue_mode was called with (lanes.size() may be smaller)
    hsa_executable_t exe{};
    hsa_code_object_reader_t reader{};
    bool have_exe = false, have_reader = false;
    std::vector<char> image;          // the gfx950 code object (kept alive for the executable)
    uint64_t kernel_object = 0;
    int v_integ = -1, v_params = -1, v_rmode = -1, v_split = -1, v_prep = -1;   // the instantiation kernel_object belongs to
    uint32_t kernarg_size = 0, group_size = 0, private_size = 0;
    unsigned block = 0;
    size_t stride = 0, slots = 0;
    unsigned *d_owner = nullptr;      // [tiles] 32-bit words, agent-scope atomics only
    unsigned *d_err = nullptr;        // the placement guard's error word: pinned, coherent HOST memory (d_err = its device address) -- a
    volatile unsigned *h_err = nullptr;   // misplaced workgroup sets it with a system-scope atomic, and every later submission sees it
                                          // without a synchronisation: a stream-ordered loop that never drains still fails within a few steps
    bool kernargs_on_device = false;  // kernarg ring in BAR-mapped device memory (else: host memory, correct but slow)
    bool dirty = false;               // packets enqueued since the last drain
    bool hip_dirty = true;            // the handle did HIP-side work since the last packet
    // QS_ORDER_STREAM
    bool stream_ordered = false;
    void *fwd_ptr = nullptr;          // value word of the forward signal (HIP signal memory)
    hsa_signal_t fwd{};
    uint64_t fwd_seq = 0;             // submissions so far
    int dbg_shift = 0;                // one-shot: the next step packet runs with StepArgs::dbg_shift (placement-guard test)
};

namespace {

#define HSA_TRY(expr)                                                                                   \
    do {                                                                                                \
        hsa_status_t s_ = (expr);                                                                       \
        if (s_ != HSA_STATUS_SUCCESS) {                                                                 \
            const char *m_ = nullptr;                                                                   \
            hsa_status_string(s_, &m_);                                                                 \
            return fail(QS_ERR_HIP, "%s failed: %s", #expr, m_ ? m_ : "unknown HSA status");           \
        }                                                                                               \
    } while (0)

constexpr int64_t kRevStart = (int64_t)1 << 40;

struct AgentPick {
    uint32_t want_bdf, want_domain;
    int want_index, seen;
    hsa_agent_t gpu, cpu;
    bool have_gpu, have_cpu;
};

hsa_status_t chain_agent_cb(hsa_agent_t a, void *data)
{
    AgentPick *p = (AgentPick *)data;
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_CPU && !p->have_cpu) { p->cpu = a; p->have_cpu = true; }
    if (t == HSA_DEVICE_TYPE_GPU) {
        uint32_t bdf = 0, dom = 0;
        const bool ok = hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS &&
                        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom) == HSA_STATUS_SUCCESS;
        // the agent must be THE device of the handle: matched by PCI domain + bus/device/function (BDFID = bus << 8 | device << 3 |
        // function); by ordinal only when HIP cannot name the device's PCI address
        if (p->want_bdf != 0xffffffffu ? (ok && bdf == p->want_bdf && dom == p->want_domain) : p->seen == p->want_index) {
            p->gpu = a;
            p->have_gpu = true;
        }
        ++p->seen;
    }
    return HSA_STATUS_SUCCESS;
}

// device-local memory the host may write through the PCIe BAR: where HIP itself keeps kernel arguments on this platform
// (kernargs in host memory would cost every workgroup a PCIe read: 44 us per step instead of 5)
struct DevPoolPick {
    hsa_agent_t cpu;
    hsa_amd_memory_pool_t pool;
    bool found;
};
hsa_status_t chain_device_pool_cb(hsa_amd_memory_pool_t pool, void *data)
{
    DevPoolPick *p = (DevPoolPick *)data;
    hsa_amd_segment_t seg;
    if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL)
        return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    bool alloc = false;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (!alloc || !(flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED)) return HSA_STATUS_SUCCESS;
    hsa_amd_memory_pool_access_t acc = HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
    hsa_amd_agent_memory_pool_get_info(p->cpu, pool, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &acc);
    if (acc == HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED) return HSA_STATUS_SUCCESS;
    p->pool = pool;
    p->found = true;
    return HSA_STATUS_INFO_BREAK;
}

hsa_status_t chain_kernarg_pool_cb(hsa_amd_memory_pool_t pool, void *data)
{
    hsa_amd_segment_t seg;
    if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL)
        return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) { *(hsa_amd_memory_pool_t *)data = pool; return HSA_STATUS_INFO_BREAK; }
    return HSA_STATUS_SUCCESS;
}

// the gfx950 code object of THIS library: the .hip_fatbin section of the shared object the code runs from holds a clang
// offload bundle; its amdgcn entry is the ELF that HIP itself loads
int chain_read_code_object(std::vector<char> &out)
{
    Dl_info di;
    if (!dladdr((void *)&chain_read_code_object, &di) || !di.dli_fname) return fail(QS_ERR_HIP, "queue mode: cannot locate the library file");
    const int fd = open(di.dli_fname, O_RDONLY);
    if (fd < 0) return fail(QS_ERR_HIP, "queue mode: cannot open %s", di.dli_fname);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return fail(QS_ERR_HIP, "queue mode: fstat failed"); }
    const char *base = (const char *)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (base == MAP_FAILED) return fail(QS_ERR_HIP, "queue mode: mmap failed");
    int rc = fail(QS_ERR_HIP, "queue mode: no gfx950 code object in %s", di.dli_fname);
    const Elf64_Ehdr *eh = (const Elf64_Ehdr *)base;
    const Elf64_Shdr *sh = (const Elf64_Shdr *)(base + eh->e_shoff);
    const char *names = base + sh[eh->e_shstrndx].sh_offset;
    for (int i = 0; i < eh->e_shnum; ++i) {
        if (strcmp(names + sh[i].sh_name, ".hip_fatbin") != 0) continue;
        const char *fb = base + sh[i].sh_offset;
        const char magic[] = "__CLANG_OFFLOAD_BUNDLE__";
        if (sh[i].sh_size < 32 || memcmp(fb, magic, 24) != 0) break;
        uint64_t n;
        memcpy(&n, fb + 24, 8);
        const char *q = fb + 32;
        for (uint64_t k = 0; k < n; ++k) {
            uint64_t off, size, tl;
            memcpy(&off, q, 8); memcpy(&size, q + 8, 8); memcpy(&tl, q + 16, 8);
            const char *triple = q + 24;
            q += 24 + tl;
            if (tl >= 6 && memmem(triple, tl, "amdgcn", 6) && memmem(triple, tl, "gfx950", 6) && off + size <= sh[i].sh_size) {
                out.assign(fb + off, fb + off + size);
                rc = QS_OK;
            }
        }
        break;
    }
    munmap((void *)base, (size_t)st.st_size);
    return rc;
}

void chain_close(QsEnv *e)
{
    QsChain *c = e->chain;
    if (!c) return;
    // nothing may be left waiting on a hand-shake value that will never come
    if (c->fwd.handle) hsa_signal_store_screlease(c->fwd, INT64_MAX);
    for (QsChainLane &L : c->lanes) {
        if (L.queue) hsa_queue_destroy(L.queue);
        if (L.kernargs) hsa_amd_memory_pool_free(L.kernargs);
        if (L.done.handle) hsa_signal_destroy(L.done);
        if (L.rev_ptr) (void)hipFree(L.rev_ptr);
    }
    if (c->fwd_ptr) (void)hipFree(c->fwd_ptr);
    if (c->have_exe) hsa_executable_destroy(c->exe);
    if (c->have_reader) hsa_code_object_reader_destroy(c->reader);
    if (c->d_owner) (void)hipFree(c->d_owner);
    if (c->h_err) (void)hipHostFree((void *)c->h_err);
    delete c;
    e->chain = nullptr;
}

// the step-kernel instantiation launch_env_on would pick for the handle AS IT IS NOW (qs_set_params / qs_set_init_state after
// qs_set_queue_mode change it): (re-)resolved against the loaded executable whenever it differs from the one in use
int chain_resolve_kernel(QsEnv *e)
{
    QsChain *c = e->chain;
    const int integ = e->cfg.integrator == QS_INTEG_FROZEN ? 0 : 1;
    const int rmode = e->init ? 3 : e->cfg.randomise;
    const int params = (rmode == 2 || e->per_env_params) ? 1 : 0;
    static const int forced = getenv("QS_SPLIT") ? atoi(getenv("QS_SPLIT")) : -1;
    const int split = (forced >= 0 ? forced != 0 : e->n <= kSplitMaxEnvs) ? 1 : 0;
    int64_t lane_tiles = 0;
    for (const QsChainLane &L : c->lanes) lane_tiles = std::max<int64_t>(lane_tiles, L.tile_end - L.tile0);
    if (c->lanes.empty()) lane_tiles = e->tiles;          // chain_open resolves once before the lanes exist: re-resolved at the first step
    const int prep = split ? prep_for(rmode, lane_tiles) : 0;
    if (c->kernel_object && integ == c->v_integ && rmode == c->v_rmode && params == c->v_params && split == c->v_split && prep == c->v_prep) return QS_OK;
    char sym[160];
    if (split) snprintf(sym, sizeof sym, "_ZN12_GLOBAL__N_111k_env_splitILi%dELb%dELi%dELi%dEEEvNS_8StepArgsE.kd", integ, params, rmode, prep);
    else snprintf(sym, sizeof sym, "_ZN12_GLOBAL__N_15k_envILi%dELb%dELi%dEEEvNS_8StepArgsE.kd", integ, params, rmode);
    hsa_executable_symbol_t ks;
    uint64_t ko = 0;
    uint32_t ka = 0, gs = 0, ps = 0;
    HSA_TRY(hsa_executable_get_symbol_by_name(c->exe, sym, &c->gpu, &ks));
    HSA_TRY(hsa_executable_symbol_get_info(ks, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &ko));
    HSA_TRY(hsa_executable_symbol_get_info(ks, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &ka));
    HSA_TRY(hsa_executable_symbol_get_info(ks, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &gs));
    HSA_TRY(hsa_executable_symbol_get_info(ks, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &ps));
    if (ka < sizeof(StepArgs)) return fail(QS_ERR_HIP, "qs_set_queue_mode: kernel argument block is %u B, StepArgs %zu B", ka, sizeof(StepArgs));
    if (c->stride && (((size_t)ka + 255) & ~size_t(255)) > c->stride)
        return fail(QS_ERR_HIP, "queue mode: kernel argument block of %s (%u B) exceeds the ring's slot size", sym, ka);
    c->kernel_object = ko; c->kernarg_size = ka; c->group_size = gs; c->private_size = ps;
    c->block = split ? split_waves(rmode, prep) * kTile : kBlock;
    c->v_integ = integ; c->v_rmode = rmode; c->v_params = params; c->v_split = split; c->v_prep = prep;
    return QS_OK;
}

// HIP "signal memory": an HSA signal created by HIP, of which it hands out the address of the value word.  The handle our
// own AQL packets need is the amd_signal_t around that word (amd_hsa_signal.h: value at offset 8, 64-byte aligned).
int chain_alloc_hip_signal(void **value_ptr, hsa_signal_t *handle, int64_t initial)
{
    HIP_TRY(hipExtMallocWithFlags(value_ptr, 8, hipMallocSignalMemory));
    const uintptr_t h = (uintptr_t)*value_ptr - offsetof(amd_signal_t, value);
    if (h & (AMD_SIGNAL_ALIGN_BYTES - 1)) return fail(QS_ERR_HIP, "queue mode: HIP signal memory is not the value word of an amd_signal_t");
    handle->handle = (uint64_t)h;
    if (((amd_signal_t *)h)->kind != AMD_SIGNAL_KIND_USER) return fail(QS_ERR_HIP, "queue mode: HIP signal memory is not a user signal");
    hsa_signal_store_screlease(*handle, initial);
    if (hsa_signal_load_scacquire(*handle) != initial) return fail(QS_ERR_HIP, "queue mode: HIP signal memory does not behave like an HSA signal");
    return QS_OK;
}

bool chain_can_stream_order(QsEnv *e)
{
    int can = 0;
    return hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, e->cfg.device) == hipSuccess && can != 0;
}

int chain_enable_stream_order(QsEnv *e)
{
    QsChain *c = e->chain;
    if (!c->fwd_ptr) {
        int r = chain_alloc_hip_signal(&c->fwd_ptr, &c->fwd, 0);
        if (r) return r;
        c->fwd_seq = 0;
    }
    for (QsChainLane &L : c->lanes) {
        if (L.rev_ptr) continue;
        int r = chain_alloc_hip_signal(&L.rev_ptr, &L.rev, kRevStart);
        if (r) return r;
        L.rev_value = kRevStart;
    }
    c->stream_ordered = true;
    return QS_OK;
}

int chain_open(QsEnv *e, int nq)
{
    if (e->cfg.kind == QS_KIND_HOVERING_V0) return fail(QS_ERR_INVALID, "qs_set_queue_mode: docking envs only");
    if (e->cfg.io_space != QS_IO_DEVICE) return fail(QS_ERR_INVALID, "qs_set_queue_mode: device buffers only");
    QsChain *c = new (std::nothrow) QsChain();
    if (!c) return fail(QS_ERR_NOMEM, "qs_set_queue_mode: out of host memory");
    e->chain = c;
    auto body = [&]() -> int {
        HSA_TRY(hsa_init());
        char bus[32] = "";
        AgentPick pick{0xffffffffu, 0, e->cfg.device, 0, {}, {}, false, false};
        unsigned dom = 0, b = 0, d = 0, f = 0;
        if (hipDeviceGetPCIBusId(bus, sizeof bus, e->cfg.device) == hipSuccess && sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) == 4) {
            pick.want_bdf = (b << 8) | (d << 3) | f;
            pick.want_domain = dom;
        }
        HSA_TRY(hsa_iterate_agents(chain_agent_cb, &pick));
        if (!pick.have_gpu || !pick.have_cpu) return fail(QS_ERR_HIP, "qs_set_queue_mode: no HSA agent for device %d (%s)", e->cfg.device, bus);
        c->gpu = pick.gpu; c->cpu = pick.cpu;
        hsa_status_t ps = hsa_amd_agent_iterate_memory_pools(c->cpu, chain_kernarg_pool_cb, &c->kernarg_pool);
        if (ps != HSA_STATUS_INFO_BREAK) return fail(QS_ERR_HIP, "qs_set_queue_mode: no kernarg memory pool");
        int r = chain_read_code_object(c->image);
        if (r) return r;
        HSA_TRY(hsa_code_object_reader_create_from_memory(c->image.data(), c->image.size(), &c->reader));
        c->have_reader = true;
        HSA_TRY(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &c->exe));
        c->have_exe = true;
        HSA_TRY(hsa_executable_load_agent_code_object(c->exe, c->gpu, c->reader, nullptr, nullptr));
        HSA_TRY(hsa_executable_freeze(c->exe, nullptr));
        r = chain_resolve_kernel(e);
        if (r) return r;
        c->stride = (((size_t)c->kernarg_size + 255) & ~size_t(255)) + 256;   // room for any instantiation's hidden arguments
        // a small ring: a slot is rewritten only after its packet ran, and recently used kernarg lines are still in the caches
        // (4 096 slots: 5.43 us per step, 256: 5.27, 64 and 16: 5.24, 4: host-bound; profiles/r02/ab_experiments.txt, section E)
        c->slots = getenv("QS_CHAIN_SLOTS") ? (size_t)atoi(getenv("QS_CHAIN_SLOTS")) : 64;
        if (c->slots < 2 || c->slots > 4096) c->slots = 64;
        DevPoolPick dp{c->cpu, {}, false};
        (void)hsa_amd_agent_iterate_memory_pools(c->gpu, chain_device_pool_cb, &dp);
        c->kernargs_on_device = dp.found;
        // nq queues, each stepping a contiguous range of tiles (whole multiples of 8 tiles where possible: one per XCD)
        std::vector<int64_t> cut{0};
        for (int q = 1; q <= nq; ++q) {
            int64_t t1 = q == nq ? e->tiles : ((e->tiles * q / nq) + 7) / 8 * 8;
            if (t1 > e->tiles) t1 = e->tiles;
            if (t1 > cut.back()) cut.push_back(t1);        // small handles: fewer, non-empty lanes
        }
        c->lanes.resize(cut.size() - 1);
        for (size_t q = 0; q + 1 < cut.size(); ++q) {
            QsChainLane &L = c->lanes[q];
            L.tile0 = cut[q]; L.tile_end = cut[q + 1];
            HSA_TRY(hsa_queue_create(c->gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &L.queue));
            if (dp.found) {
                HSA_TRY(hsa_amd_memory_pool_allocate(dp.pool, c->stride * c->slots, 0, (void **)&L.kernargs));
                HSA_TRY(hsa_amd_agents_allow_access(1, &c->cpu, nullptr, L.kernargs));
            } else {
                HSA_TRY(hsa_amd_memory_pool_allocate(c->kernarg_pool, c->stride * c->slots, 0, (void **)&L.kernargs));
                HSA_TRY(hsa_amd_agents_allow_access(1, &c->gpu, nullptr, L.kernargs));
            }
            memset(L.kernargs, 0, c->stride * c->slots);   // the hidden arguments behind StepArgs are never read: zeros
            L.slot_qidx.assign(c->slots, 0);
            HSA_TRY(hsa_signal_create(0, 0, nullptr, &L.done));
        }
        HIP_TRY(hipMalloc((void **)&c->d_owner, (size_t)e->tiles * sizeof(unsigned)));
        HIP_TRY(hipHostMalloc((void **)&c->h_err, 64, hipHostMallocMapped | hipHostMallocCoherent));
        *c->h_err = 0;
        HIP_TRY(hipHostGetDevicePointer((void **)&c->d_err, (void *)c->h_err, 0));
        // stream-ordered hand-shake: needs HIP's stream memory operations and its signal memory; without them the mode
        // stays host-ordered (round 2's contract)
        const char *ord = getenv("QS_CHAIN_ORDER");            // "host": start with round 2's contract (A/B runs)
        if (!(ord && ord[0] == 'h') && chain_can_stream_order(e)) {
            r = chain_enable_stream_order(e);
            if (r) return r;
        }
        return QS_OK;
    };
    int rc;
    try { rc = body(); }                                 // no C++ exception may cross the C ABI
    catch (...) { rc = fail(QS_ERR_NOMEM, "qs_set_queue_mode: out of host memory"); }
    if (rc != QS_OK) chain_close(e);
    return rc;
}

// one AQL packet behind everything enqueued before it on lane L (barrier bit): kind 0 the step kernel, 1 a barrier-AND packet
// (drain), 2 an AMD barrier-value packet that holds the lane until `wait_sig` >= wait_value (the caller's stream is ready)
enum { PKT_STEP = 0, PKT_BARRIER = 1, PKT_WAIT_VALUE = 2 };
uint64_t chain_write_packet(QsChain *c, QsChainLane &L, int kind, const void *kernarg, unsigned grid, int acquire, int release,
                            hsa_signal_t completion, hsa_signal_t wait_sig = hsa_signal_t{0}, int64_t wait_value = 0)
{
    const uint64_t idx = hsa_queue_add_write_index_relaxed(L.queue, 1);
    while (idx - hsa_queue_load_read_index_scacquire(L.queue) >= L.queue->size) __builtin_ia32_pause();
    void *slot = (char *)L.queue->base_address + (idx & (L.queue->size - 1)) * 64;
    uint32_t word0;                                 // header (16 bits) + the 16 bits behind it, published by ONE 32-bit store
    const uint16_t fences = (1 << HSA_PACKET_HEADER_BARRIER) | (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    if (kind == PKT_BARRIER) {
        hsa_barrier_and_packet_t *p = (hsa_barrier_and_packet_t *)slot;
        memset((char *)p + 4, 0, 60);
        p->completion_signal = completion;
        word0 = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | fences);
    } else if (kind == PKT_WAIT_VALUE) {
        hsa_amd_barrier_value_packet_t *p = (hsa_amd_barrier_value_packet_t *)slot;
        memset((char *)p + 4, 0, 60);
        p->signal = wait_sig;
        p->value = wait_value;
        p->mask = -1;
        p->cond = HSA_SIGNAL_CONDITION_GTE;
        p->completion_signal = completion;
        word0 = (uint16_t)((HSA_PACKET_TYPE_VENDOR_SPECIFIC << HSA_PACKET_HEADER_TYPE) | fences) |
                ((uint32_t)HSA_AMD_PACKET_TYPE_BARRIER_VALUE << 16);
    } else {
        hsa_kernel_dispatch_packet_t *p = (hsa_kernel_dispatch_packet_t *)slot;
        p->workgroup_size_x = (uint16_t)c->block; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
        p->reserved0 = 0;
        p->grid_size_x = grid; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = c->private_size;
        p->group_segment_size = c->group_size;
        p->kernel_object = c->kernel_object;
        p->kernarg_address = (void *)kernarg;
        p->reserved2 = 0;
        p->completion_signal = completion;
        word0 = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | fences) |
                ((uint32_t)(1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS) << 16);
    }
    __atomic_store_n((uint32_t *)slot, word0, __ATOMIC_RELEASE);
    hsa_signal_store_screlease(L.queue->doorbell_signal, (hsa_signal_value_t)idx);
    return idx;
}

// every packet has run and what it wrote is visible to the whole system (host wait); reports a misplaced tile
int chain_drain(QsEnv *e)
{
    QsChain *c = e->chain;
    if (!c || !c->dirty) return QS_OK;
    for (QsChainLane &L : c->lanes) {
        hsa_signal_store_relaxed(L.done, 1);
        chain_write_packet(c, L, PKT_BARRIER, nullptr, 0, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_SYSTEM, L.done);
    }
    for (QsChainLane &L : c->lanes)
        while (hsa_signal_wait_scacquire(L.done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) {}
    c->dirty = false;
    if (*c->h_err) {
        *c->h_err = 0;
        return fail(QS_ERR_HIP, "queue mode: a workgroup ran on another XCD than the one holding its tile; the steps since the last "
                                "synchronisation are invalid (this placement is not promised by HIP: use qs_set_queue_mode(env, 0))");
    }
    return QS_OK;
}

// T consecutive steps (T kernarg blocks: steps[t] differ in their I/O pointers only) on every lane, behind ONE hand-shake with
// the handle's stream when the chain is stream-ordered
int chain_submit(QsEnv *e, const StepArgs *steps, int64_t T)
{
    QsChain *c = e->chain;
    if (*c->h_err) {
        // a workgroup of an EARLIER step found its tile on another XCD (the word is host memory: no synchronisation needed to see
        // it).  Reported here as well as at the next drain, so that a loop of nothing but steps cannot run on unnoticed; the flag
        // stays set until a draining call has reported it and re-armed the handle.
        return fail(QS_ERR_HIP, "queue mode: a workgroup ran on another XCD than the one holding its tile; the steps since the last "
                                "synchronisation are invalid (call qs_sync, then continue or use qs_set_queue_mode(env, 0))");
    }
    int r = chain_resolve_kernel(e);               // qs_set_params / qs_set_init_state since the last step?
    if (r) return r;
    if (c->stream_ordered) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(e->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            return fail(QS_ERR_INVALID, "queue mode: a private-queue step cannot be captured into a hipGraph (use qs_set_queue_mode(env, 0))");
    }
    if (c->hip_dirty) {
        // HIP-side work of the handle (reset, set_state, ...) must have finished, and no tile has an owning XCD yet
        HIP_TRY(hipMemsetAsync(c->d_owner, 0xff, (size_t)e->tiles * sizeof(unsigned), e->stream));
        if (!c->stream_ordered) HIP_TRY(hipStreamSynchronize(e->stream));    // stream-ordered: the write-value below is behind it
        c->hip_dirty = false;
    }
    const size_t nl = c->lanes.size();
    if (c->stream_ordered) {
        ++c->fwd_seq;
        HIP_TRY(hipStreamWriteValue64(e->stream, c->fwd_ptr, c->fwd_seq, 0));
        for (QsChainLane &L : c->lanes)
            chain_write_packet(c, L, PKT_WAIT_VALUE, nullptr, 0, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, hsa_signal_t{0}, c->fwd,
                               (int64_t)c->fwd_seq);
    }
    for (int64_t t = 0; t < T; ++t) {
        StepArgs A = steps[t];
        A.owner = c->d_owner;
        A.err = c->d_err;
        A.dbg_shift = c->dbg_shift;
        c->dbg_shift = 0;
        // all lanes' kernarg blocks first, ONE read-back behind them, then the packets: the read-back is a PCIe round trip
        char *ka[8];
        size_t slot[8];
        for (size_t q = 0; q < nl; ++q) {
            QsChainLane &L = c->lanes[q];
            A.tile0 = L.tile0; A.tile_end = L.tile_end;
            // the slot about to be rewritten belongs to step `issued - slots`, queue packet p: that kernel has FINISHED once the
            // packet behind it has been taken off the queue (every packet carries the barrier bit): read index past p + 1
            slot[q] = L.issued % c->slots;
            if (L.issued >= c->slots)
                while (hsa_queue_load_read_index_scacquire(L.queue) < L.slot_qidx[slot[q]] + 2) __builtin_ia32_pause();
            ka[q] = L.kernargs + slot[q] * c->stride;
            memcpy(ka[q], &A, sizeof A);
        }
        if (c->kernargs_on_device) {
            // posted writes through the BAR: reading the last word back makes sure they have landed before a doorbell rings
            __builtin_ia32_sfence();
            (void)*(volatile uint32_t *)(ka[nl - 1] + sizeof A - sizeof(uint32_t));
        }
        const bool last = t + 1 == T;
        for (size_t q = 0; q < nl; ++q) {
            QsChainLane &L = c->lanes[q];
            const int64_t tiles = L.tile_end - L.tile0;
            const unsigned grid = c->v_split ? (unsigned)(tiles * c->block)
                                                        : (unsigned)(((tiles + kBlock / kTile - 1) / (kBlock / kTile)) * kBlock);
            // stream-ordered: the LAST packet of the submission publishes -- agent-scope release (the outputs of all T steps leave
            // the L2s; the state lines are written back too but stay valid where they are) and the completion signal the
            // caller's stream waits for.  (Write-through `sc1` output stores on every step instead of this release were built
            // and measured: 6.90 against 5.74 us per step in a 600-step roll-out, and they cost the ordinary launches 0.9 us
            // through the store code they displaced; profiles/r03/ab_experiments.txt section I.)
            const bool sig = c->stream_ordered && last;
            L.slot_qidx[slot[q]] = chain_write_packet(c, L, PKT_STEP, ka[q], grid, HSA_FENCE_SCOPE_AGENT,
                                                      sig ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE, sig ? L.rev : hsa_signal_t{0});
            ++L.issued;
            if (sig) --L.rev_value;
        }
    }
    c->dirty = true;
    if (c->stream_ordered)
        for (QsChainLane &L : c->lanes)
            HIP_TRY(hipStreamWaitValue64(e->stream, L.rev_ptr, (uint64_t)L.rev_value, hipStreamWaitValueEq, ~0ull));
    return QS_OK;
}

int chain_step(QsEnv *e, const StepArgs &A) { return chain_submit(e, &A, 1); }

}  // namespace
