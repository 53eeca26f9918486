// env_groups.hpp -- env groups (qs_set_groups): tile ranges of a handle stepped on streams of their own, optional launcher threads
// A fragment of quadsim_hip.hip (ONE translation unit: the kernels' mangled names, which the private-queue code resolves
// in the code object, live in that unit's anonymous namespace); included there at a fixed position, nowhere else.
#pragma once

// ---- env groups -----------------------------------------------------------------------------------------------------
// A handle's tiles can be partitioned into G contiguous groups, each stepped on its OWN stream (EnvPool-style: a trainer
// runs the policy of one group while the others step).  Envs never interact, so a group launch is the ordinary step kernel
// over a tile range: results are bit-identical to the single launch.  What the groups buy is overlap: at 65 536 envs one
// step is ~5 us of kernel plus a ~1.8 us dependent-kernel boundary (MI355X_MICROARCH.md, price list, "boundary"); with two
// chains in flight one group's boundary and wave ramp hide under the other group's compute.  Two launches per step would
// make ONE host thread the bottleneck (~2.5-3 us per launch), so each group may get its own launcher thread: the API
// thread posts a launch record into a single-producer ring and returns; the group's thread issues it on the group's stream.
struct QsGroup {
    enum { kRing = 64 };
    enum ReqType { REQ_LAUNCH = 0, REQ_WAIT_EVENT = 1 };
    struct Req {
        int type;
        StepArgs A;
        hipEvent_t ev;
    };
    QsEnv *env = nullptr;
    int index = 0;
    int64_t tile0 = 0, tile_end = 0, env0 = 0, env_end = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t done_ev = nullptr;
    // launcher thread (optional)
    bool threaded = false;
    std::thread th;
    Req ring[kRing];
    std::atomic<uint64_t> head{0}, tail{0};     // posted / issued
    std::atomic<int> stop{0}, sleeping{0}, err{0};
    std::mutex m;
    std::condition_variable cv;
};

namespace {

int group_execute(QsGroup *g, const QsGroup::Req &r)
{
    if (r.type == QsGroup::REQ_LAUNCH) return launch_env_on(g->env, r.A, g->stream);
    HIP_TRY(hipStreamWaitEvent(g->stream, r.ev, 0));
    return QS_OK;
}

void group_worker(QsGroup *g)
{
    (void)hipSetDevice(g->env->cfg.device);
    for (;;) {
        const uint64_t t = g->tail.load(std::memory_order_relaxed);
        int spins = 0;
        while (g->head.load(std::memory_order_acquire) == t) {
            if (g->stop.load(std::memory_order_acquire)) return;
            if (++spins < 40000) { __builtin_ia32_pause(); continue; }
            // idle for ~100 us: sleep until the API thread posts again (it checks `sleeping` after publishing)
            std::unique_lock<std::mutex> lk(g->m);
            g->sleeping.store(1, std::memory_order_seq_cst);
            if (g->head.load(std::memory_order_seq_cst) == t && !g->stop.load())
                g->cv.wait_for(lk, std::chrono::milliseconds(50));
            g->sleeping.store(0, std::memory_order_seq_cst);
            spins = 0;
        }
        const int rc = group_execute(g, g->ring[t % QsGroup::kRing]);
        if (rc != QS_OK) { int z = 0; g->err.compare_exchange_strong(z, rc); }
        g->tail.store(t + 1, std::memory_order_release);
    }
}

int group_post(QsGroup *g, const QsGroup::Req &r)
{
    if (!g->threaded) return group_execute(g, r);
    const uint64_t h = g->head.load(std::memory_order_relaxed);
    while (h - g->tail.load(std::memory_order_acquire) >= QsGroup::kRing) __builtin_ia32_pause();
    g->ring[h % QsGroup::kRing] = r;
    g->head.store(h + 1, std::memory_order_seq_cst);
    if (g->sleeping.load(std::memory_order_seq_cst)) { std::lock_guard<std::mutex> lk(g->m); g->cv.notify_one(); }
    return QS_OK;
}

// every record posted to group g has been issued to its stream
void group_wait_issued(QsGroup *g)
{
    if (!g->threaded) return;
    const uint64_t h = g->head.load(std::memory_order_relaxed);
    while (g->tail.load(std::memory_order_acquire) != h) __builtin_ia32_pause();
}

// every posted record has been issued to its stream
int groups_drain(QsEnv *e)
{
    int rc = QS_OK;
    for (QsGroup *g : e->groups) {
        if (g->threaded) {
            const uint64_t h = g->head.load(std::memory_order_relaxed);
            while (g->tail.load(std::memory_order_acquire) != h) __builtin_ia32_pause();
        }
        const int ge = g->err.exchange(0);
        if (ge != QS_OK && rc == QS_OK) rc = fail(ge, "a group launcher thread reported error %d (group %d)", ge, g->index);
    }
    return rc;
}

// group streams wait for everything enqueued so far on the main stream
int groups_fork(QsEnv *e)
{
    if (e->groups.empty()) return QS_OK;
    HIP_TRY(hipEventRecord(e->fork_ev, e->stream));
    for (QsGroup *g : e->groups) {
        QsGroup::Req r;
        r.type = QsGroup::REQ_WAIT_EVENT;
        r.ev = e->fork_ev;
        int rc = group_post(g, r);
        if (rc) return rc;
    }
    // a re-record of fork_ev must not overtake a wait that has not been issued yet
    int rc = groups_drain(e);
    e->main_dirty = false;
    return rc;
}

// the main stream waits for everything enqueued so far on the group streams
int groups_join(QsEnv *e)
{
    if (e->groups.empty()) return QS_OK;
    int rc = groups_drain(e);
    if (rc) return rc;
    for (QsGroup *g : e->groups) {
        HIP_TRY(hipEventRecord(g->done_ev, g->stream));
        HIP_TRY(hipStreamWaitEvent(e->stream, g->done_ev, 0));
    }
    e->groups_dirty = false;
    return QS_OK;
}

void groups_destroy(QsEnv *e)
{
    for (QsGroup *g : e->groups) {
        if (g->threaded) {
            g->stop.store(1, std::memory_order_release);
            { std::lock_guard<std::mutex> lk(g->m); g->cv.notify_one(); }
            if (g->th.joinable()) g->th.join();
        }
        if (g->stream) (void)hipStreamSynchronize(g->stream);
        if (g->done_ev) (void)hipEventDestroy(g->done_ev);
        if (g->own_stream && g->stream) (void)hipStreamDestroy(g->stream);
        delete g;
    }
    e->groups.clear();
    if (e->fork_ev) { (void)hipEventDestroy(e->fork_ev); e->fork_ev = nullptr; }
    e->groups_dirty = false;
    e->main_dirty = true;
}

}  // namespace
