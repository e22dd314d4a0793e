// Context, workspace pool, profiling scopes, deterministic reduction finish, communicator glue.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "common.h"
#include "ops.h"

namespace bk {

int set_error(bk_ctx* ctx, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else fprintf(stderr, "bkhip: %s\n", buf);
    return -1;
}

// ------------------------------------------------------------------ workspace pool
int ws_get(bk_ctx* ctx, size_t n, double** out) {
    if (n == 0) n = 1;
    auto it = ctx->pool_free.find(n);
    if (it != ctx->pool_free.end()) {
        *out = it->second;
        ctx->pool_free.erase(it);
        return 0;
    }
    double* p = nullptr;
    hipError_t e = hipMalloc(&p, n * sizeof(double));
    if (e != hipSuccess) {
        // release cached buffers and retry once
        for (auto& kv : ctx->pool_free) { (void)hipFree(kv.second); ctx->pool_all.erase(kv.second); }
        ctx->pool_free.clear();
        e = hipMalloc(&p, n * sizeof(double));
        if (e != hipSuccess)
            return set_error(ctx, "hipMalloc of %zu doubles failed: %s", n, hipGetErrorString(e));
    }
    ctx->pool_all[p] = n;
    *out = p;
    return 0;
}

void ws_put(bk_ctx* ctx, double* p) {
    if (!p) return;
    auto it = ctx->pool_all.find(p);
    if (it == ctx->pool_all.end()) return;
    ctx->pool_free.insert({it->second, p});
}

// ------------------------------------------------------------------ profiling
static hipEvent_t get_event(bk_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(bk_ctx* c, const char* name, double alg_bytes) : ctx(c) {
    if (!ctx->prof) return;
    e = &ctx->prof_entries[name];
    e->calls += 1;
    e->bytes += alg_bytes;
    start = get_event(ctx);
    stop = get_event(ctx);
    (void)hipEventRecord(start, ctx->stream);
}

ProfScope::~ProfScope() {
    if (!e) return;
    (void)hipEventRecord(stop, ctx->stream);
    e->pending.push_back({start, stop});
}

static void prof_resolve(bk_ctx* ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->prof_entries) {
        for (auto& pr : kv.second.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) kv.second.ms += ms;
            ctx->event_pool.push_back(pr.first);
            ctx->event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

// ------------------------------------------------------------------ reductions
__global__ void __launch_bounds__(256) reduce_stage2_kernel(const double* __restrict__ partials, int nblocks,
                                                            int nvals, int op, double* __restrict__ out) {
    const int v = blockIdx.x;
    double acc = (op == 0) ? 0.0 : -1.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        const double x = partials[(size_t)b * nvals + v];
        acc = (op == 0) ? acc + x : fmax(acc, x);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(acc, off, 64);
        acc = (op == 0) ? acc + o : fmax(acc, o);
    }
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sm[0];
        for (int w = 1; w < 4; ++w) r = (op == 0) ? r + sm[w] : fmax(r, sm[w]);
        out[v] = r;
    }
}

}  // namespace bk

// ------------------------------------------------------------------ host-staged communicator: in-stream collectives
// The callbacks of bk_ctx_create_hostcomm work on HOST buffers.  To make the ranks of a host-staged context run the very code
// RCCL ranks run -- collectives ENQUEUED in the stream between kernels, no host synchronisation around them: device-resident
// Arnoldi chunks, the halo exchange on its own stream under the interior z-chunks, two lanes -- every collective becomes a
// stream-ordered hand-over to a PROXY THREAD owned by the context (the role RCCL's own proxy threads play):
//   stream:  copy the operand to pinned host memory -> post kernel: ready[slot] = seq -> wait kernel: spins (s_sleep) on
//            done[slot] == seq -> copy the result back
//   proxy:   takes the operations in the order the host enqueued them (identical on every rank), waits for ready[slot],
//            runs the callback, sets done[slot]
// The flags live in pinned, device-mapped (fine-grained) host memory.  The wait kernel gives up after "hostcomm_timeout_s"
// (30 s) and raises the error flag instead of hanging the device; a failing callback raises it too; ctx_sync() reports it.
// The callbacks therefore run on a library-owned thread (include/bkhip.h says so).
namespace {

constexpr int kProxySlots = 1024;
constexpr long long kProxyTicksPerSecond = 100000000LL;     // wall_clock64: 100 MHz
// The wait kernel's time-out is the context option "hostcomm_timeout_s" (default 30): a callback that legitimately takes longer (a
// Python callback waiting for the GIL behind a long host-side computation) needs a larger value.  A time-out or a failed callback is
// FATAL for the context: the error flag is sticky (ctx_sync keeps reporting it), the stream has continued with stale data and the
// proxy threads of the ranks are no longer in step -- destroy the context on every rank and create a new one.

__global__ void proxy_post_kernel(unsigned long long* flag, unsigned long long seq) {
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void proxy_wait_kernel(unsigned long long* flag, unsigned long long seq, int* err, long long timeout_ticks) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
        __builtin_amdgcn_s_sleep(64);
        if (wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_store(err, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    __threadfence_system();
}

struct ProxyOp {
    unsigned long long seq = 0;
    int kind = 0;                     // 0 all-reduce, 1 halo exchange, 2 all-to-all
    int n = 0, op = 0;                // all-reduce
    size_t cnt = 0;                   // halo: doubles per face
    bool has_lo = false, has_hi = false;
    std::vector<size_t> scount, sdispl, rcount, rdispl;   // all-to-all (staging offsets)
};

}  // namespace

struct CommProxy {
    bk_ctx* ctx = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<ProxyOp> q;
    bool stop = false;
    unsigned long long seq = 0;
    unsigned long long *ready = nullptr, *done = nullptr;        // pinned host, kProxySlots each
    unsigned long long *ready_dev = nullptr, *done_dev = nullptr;
    int* err = nullptr;
    int* err_dev = nullptr;
    // INVARIANT of the three staging buffers (single instances, not per-slot): two operations that use the same buffer are never in
    // flight at once, because every use is totally ordered by stream dependencies --
    //   ar        all-reduces are only ever enqueued on the context's compute stream (reduce_finish, comm_allreduce_host, the
    //             device-resident Arnoldi step): one stream, program order;
    //   hs / hr   a halo exchange runs on comm_stream under the interior z-chunks (comm_stream waits for ev_ready of the compute stream,
    //             the compute stream waits for ev_halo before the face chunks) or in line on the compute stream (halo_overlap = 0, the
    //             fused Lanczos step): the next exchange, on either stream, is enqueued behind that wait;
    //   as / ar2  all-to-alls are only enqueued on the compute stream (the distributed preconditioner).
    // The second lane is a separate context with its own proxy and buffers.  A new call site on another stream breaks this: give it its
    // own buffers (or index the buffers by slot).
    double* ar = nullptr;                                        // all-reduce staging (kRedSlots)
    double *hs = nullptr, *hr = nullptr;                         // halo staging: [lo | hi] faces
    size_t hcap = 0;
    double *as = nullptr, *ar2 = nullptr;                        // all-to-all staging
    size_t acap = 0;

    void run() {
        for (;;) {
            ProxyOp op;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                op = q.front();
            }
            const int slot = (int)(op.seq % kProxySlots);
            int spins = 0;
            while (__atomic_load_n(&ready[slot], __ATOMIC_ACQUIRE) != op.seq) {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (stop) return;
                }
                if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(20));
                else std::this_thread::yield();
            }
            int rc = 0;
            const int me = ctx->rank, R = ctx->nranks;
            if (op.kind == 0) {
                rc = ctx->h_allreduce(ctx->h_user, ar, op.n, op.op);
            } else if (op.kind == 1) {
                // even ranks talk to the upper neighbour first, odd ranks to the lower one (deadlock-free pairing)
                for (int phase = 0; phase < 2 && rc == 0; ++phase) {
                    const bool up = ((me & 1) == 0) ? (phase == 0) : (phase == 1);
                    if (up && op.has_hi) rc = ctx->h_sendrecv(ctx->h_user, hs + op.cnt, op.cnt, me + 1, hr + op.cnt, op.cnt, me + 1);
                    else if (!up && op.has_lo) rc = ctx->h_sendrecv(ctx->h_user, hs, op.cnt, me - 1, hr, op.cnt, me - 1);
                }
            } else {
                for (int s_ = 1; s_ < R && rc == 0; ++s_) {
                    const int dst = (me + s_) % R, src = (me - s_ + R) % R;
                    rc = ctx->h_sendrecv(ctx->h_user, as + op.sdispl[dst], op.scount[dst], dst, ar2 + op.rdispl[src], op.rcount[src], src);
                }
            }
            if (rc != 0) __atomic_store_n(err, 2, __ATOMIC_RELEASE);
            __atomic_store_n(&done[slot], op.seq, __ATOMIC_RELEASE);
            {
                std::lock_guard<std::mutex> lk(mu);
                q.pop_front();
            }
        }
    }
};

namespace bk {

static int proxy_get(bk_ctx* ctx, CommProxy** out) {
    if (ctx->proxy) { *out = ctx->proxy; return 0; }
    CommProxy* p = new CommProxy();
    p->ctx = ctx;
    void* d = nullptr;
    auto pinned = [&](void** host, size_t bytes) -> hipError_t { return hipHostMalloc(host, bytes, hipHostMallocMapped | hipHostMallocCoherent); };
    hipError_t e = pinned((void**)&p->ready, sizeof(unsigned long long) * kProxySlots);
    if (e == hipSuccess) e = pinned((void**)&p->done, sizeof(unsigned long long) * kProxySlots);
    if (e == hipSuccess) e = pinned((void**)&p->err, sizeof(int) * 16);
    if (e == hipSuccess) e = pinned((void**)&p->ar, sizeof(double) * kRedSlots);
    if (e == hipSuccess) { memset(p->ready, 0xff, sizeof(unsigned long long) * kProxySlots); memset(p->done, 0xff, sizeof(unsigned long long) * kProxySlots); p->err[0] = 0; }
    if (e == hipSuccess) e = hipHostGetDevicePointer(&d, p->ready, 0);
    p->ready_dev = static_cast<unsigned long long*>(d);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&d, p->done, 0);
    p->done_dev = static_cast<unsigned long long*>(d);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&d, p->err, 0);
    p->err_dev = static_cast<int*>(d);
    if (e != hipSuccess) {
        delete p;
        return set_error(ctx, "host communicator: proxy allocation failed: %s", hipGetErrorString(e));
    }
    p->th = std::thread([p] { p->run(); });
    ctx->proxy = p;
    *out = p;
    return 0;
}

void proxy_destroy(bk_ctx* ctx) {
    CommProxy* p = ctx->proxy;
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv.notify_all();
    if (p->th.joinable()) p->th.join();
    for (void* h : {(void*)p->ready, (void*)p->done, (void*)p->err, (void*)p->ar, (void*)p->hs, (void*)p->hr, (void*)p->as, (void*)p->ar2})
        if (h) (void)hipHostFree(h);
    delete p;
    ctx->proxy = nullptr;
}

// enqueue the hand-over of `op` on `stream`: post kernel, queue entry, wait kernel (the caller copies operands before / results after)
static int proxy_submit(bk_ctx* ctx, CommProxy* p, hipStream_t stream, ProxyOp& op) {
    op.seq = p->seq++;
    const int slot = (int)(op.seq % kProxySlots);
    hipLaunchKernelGGL(proxy_post_kernel, dim3(1), dim3(1), 0, stream, p->ready_dev + slot, op.seq);
    BK_HIP(ctx, hipGetLastError());
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->q.push_back(op);
    }
    p->cv.notify_one();
    const double tmo = std::max(1.0, ctx->opt("hostcomm_timeout_s", 30.0));
    hipLaunchKernelGGL(proxy_wait_kernel, dim3(1), dim3(1), 0, stream, p->done_dev + slot, op.seq, p->err_dev,
                       (long long)(tmo * (double)kProxyTicksPerSecond));
    BK_HIP(ctx, hipGetLastError());
    return 0;
}

static int proxy_grow(bk_ctx* ctx, hipStream_t stream, double** a, double** b, size_t* cap, size_t need) {
    if (need <= *cap) return 0;
    BK_HIP(ctx, hipStreamSynchronize(stream));           // earlier operations may still use the old buffers
    if (ctx->comm_stream) BK_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));
    if (*a) (void)hipHostFree(*a);
    if (*b) (void)hipHostFree(*b);
    *a = *b = nullptr;
    *cap = 0;
    // (coherent like the flags and the all-reduce staging buffer: the proxy thread reads what a stream-ordered copy wrote and the
    // stream reads what the proxy wrote, with no host synchronisation in between -- ADVICE r4)
    BK_HIP(ctx, hipHostMalloc((void**)a, need * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    BK_HIP(ctx, hipHostMalloc((void**)b, need * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    *cap = need;
    return 0;
}

// In-stream all-reduce of a device buffer (n <= kRedSlots doubles), both communicator kinds: what an RCCL rank enqueues as
// ncclAllReduce, a host-staged rank hands to its proxy thread.  No host synchronisation.
int comm_allreduce_dev(bk_ctx* ctx, hipStream_t stream, double* dbuf, int n, int op) {
    if (ctx->comm == COMM_NONE || ctx->nranks == 1) return 0;
    if (n > kRedSlots) return set_error(ctx, "comm_allreduce_dev: n too large");
    if (ctx->comm == COMM_RCCL) {
        BK_NCCL(ctx, ncclAllReduce(dbuf, dbuf, n, ncclDouble, op == 0 ? ncclSum : ncclMax, ctx->nccl, stream));
        return 0;
    }
    CommProxy* p = nullptr;
    BK_TRY(proxy_get(ctx, &p));
    BK_HIP(ctx, hipMemcpyAsync(p->ar, dbuf, n * sizeof(double), hipMemcpyDeviceToHost, stream));
    ProxyOp o;
    o.kind = 0; o.n = n; o.op = op;
    BK_TRY(proxy_submit(ctx, p, stream, o));
    BK_HIP(ctx, hipMemcpyAsync(dbuf, p->ar, n * sizeof(double), hipMemcpyHostToDevice, stream));
    return 0;
}

// stream synchronisation + the error state of the proxied collectives (time-out of a wait kernel, failed callback)
int ctx_sync(bk_ctx* ctx) {
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->proxy) {
        const int e = __atomic_load_n(ctx->proxy->err, __ATOMIC_ACQUIRE);
        if (e == 1) return set_error(ctx, "host communicator: a collective timed out on the device (a peer rank never arrived)");
        if (e == 2) return set_error(ctx, "host communicator: a callback failed");
    }
    return 0;
}

}  // namespace bk

namespace bk {

int comm_allreduce_host(bk_ctx* ctx, double* buf, int n, int op) {
    if (ctx->comm == COMM_NONE || ctx->nranks == 1) return 0;
    // staged through the device result buffer and the in-stream all-reduce: the same path for both communicator kinds
    if (n > kRedSlots) return set_error(ctx, "comm_allreduce_host: n too large");
    BK_HIP(ctx, hipMemcpyAsync(ctx->d_red, buf, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    BK_TRY(comm_allreduce_dev(ctx, ctx->stream, ctx->d_red, n, op));
    BK_HIP(ctx, hipMemcpyAsync(buf, ctx->d_red, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return ctx_sync(ctx);
}

int reduce_finish(bk_ctx* ctx, int nblocks, int nvals, int op) {
    if (nvals > kRedSlots) return set_error(ctx, "reduce_finish: too many values");
    // (after the fact for this launch, but loud: a stage-1 grid that would not fit d_partials is a bug of its launcher)
    if ((size_t)nblocks * (size_t)nvals > kPartialDoubles) return set_error(ctx, "reduce_finish: %d x %d partial sums exceed d_partials", nblocks, nvals);
    if (ctx->nranks == 1 && ctx->h_red_dev) {
        // single rank: the second stage writes straight into the pinned, device-mapped host buffer -- no copy
        // operation, the host only waits for the stream
        hipLaunchKernelGGL(reduce_stage2_kernel, dim3(nvals), dim3(256), 0, ctx->stream, ctx->d_partials, nblocks,
                           nvals, op, ctx->h_red_dev);
        BK_HIP(ctx, hipGetLastError());
        BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    }
    hipLaunchKernelGGL(reduce_stage2_kernel, dim3(nvals), dim3(256), 0, ctx->stream, ctx->d_partials, nblocks,
                       nvals, op, ctx->d_red);
    BK_HIP(ctx, hipGetLastError());
    BK_TRY(comm_allreduce_dev(ctx, ctx->stream, ctx->d_red, nvals, op));      // in the stream, both communicator kinds
    BK_HIP(ctx, hipMemcpyAsync(ctx->h_red, ctx->d_red, nvals * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return ctx_sync(ctx);
}

// ------------------------------------------------------------------ halo exchange
// Exchange `width` boundary planes (plane = `plane` doubles) of the local slab `v` (nplanes planes)
// with the neighbour ranks: halo_lo <- last planes of rank-1, halo_hi <- first planes of rank+1.
int halo_exchange(bk_ctx* ctx, hipStream_t stream, const double* v, size_t plane, int nplanes, int width, double* halo_lo,
                  double* halo_hi) {
    if (ctx->nranks == 1) return 0;
    const int lo = ctx->rank - 1, hi = ctx->rank + 1;
    const bool has_lo = lo >= 0, has_hi = hi < ctx->nranks;
    const size_t cnt = plane * (size_t)width;
    if (nplanes < width) return set_error(ctx, "halo_exchange: slab thinner than the halo width");
    const double* send_lo = v;                                        // my first planes -> rank-1
    const double* send_hi = v + plane * (size_t)(nplanes - width);    // my last planes  -> rank+1
    if (ctx->comm == COMM_RCCL) {
        BK_NCCL(ctx, ncclGroupStart());
        if (has_lo) {
            BK_NCCL(ctx, ncclSend(send_lo, cnt, ncclDouble, lo, ctx->nccl, stream));
            BK_NCCL(ctx, ncclRecv(halo_lo, cnt, ncclDouble, lo, ctx->nccl, stream));
        }
        if (has_hi) {
            BK_NCCL(ctx, ncclSend(send_hi, cnt, ncclDouble, hi, ctx->nccl, stream));
            BK_NCCL(ctx, ncclRecv(halo_hi, cnt, ncclDouble, hi, ctx->nccl, stream));
        }
        BK_NCCL(ctx, ncclGroupEnd());
        return 0;
    }
    // host-staged communicator: stream-ordered hand-over to the proxy thread (no host synchronisation)
    CommProxy* p = nullptr;
    BK_TRY(proxy_get(ctx, &p));
    BK_TRY(proxy_grow(ctx, stream, &p->hs, &p->hr, &p->hcap, 2 * cnt));
    if (has_lo) BK_HIP(ctx, hipMemcpyAsync(p->hs, send_lo, cnt * 8, hipMemcpyDeviceToHost, stream));
    if (has_hi) BK_HIP(ctx, hipMemcpyAsync(p->hs + cnt, send_hi, cnt * 8, hipMemcpyDeviceToHost, stream));
    ProxyOp o;
    o.kind = 1; o.cnt = cnt; o.has_lo = has_lo; o.has_hi = has_hi;
    BK_TRY(proxy_submit(ctx, p, stream, o));
    if (has_lo) BK_HIP(ctx, hipMemcpyAsync(halo_lo, p->hr, cnt * 8, hipMemcpyHostToDevice, stream));
    if (has_hi) BK_HIP(ctx, hipMemcpyAsync(halo_hi, p->hr + cnt, cnt * 8, hipMemcpyHostToDevice, stream));
    return 0;
}

// ------------------------------------------------------------------ all-to-all (variable counts, doubles)
// Used by the distributed DCT preconditioner to turn z-slabs into y-slabs and back.
int comm_alltoallv(bk_ctx* ctx, const double* sendbuf, const size_t* scount, const size_t* sdispl, double* recvbuf,
                   const size_t* rcount, const size_t* rdispl) {
    const int R = ctx->nranks, me = ctx->rank;
    if (R == 1) {
        BK_HIP(ctx, hipMemcpyAsync(recvbuf + rdispl[0], sendbuf + sdispl[0], scount[0] * sizeof(double),
                                   hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    if (ctx->comm == COMM_RCCL) {
        BK_NCCL(ctx, ncclGroupStart());
        for (int p = 0; p < R; ++p) {
            if (scount[p]) BK_NCCL(ctx, ncclSend(sendbuf + sdispl[p], scount[p], ncclDouble, p, ctx->nccl, ctx->stream));
            if (rcount[p]) BK_NCCL(ctx, ncclRecv(recvbuf + rdispl[p], rcount[p], ncclDouble, p, ctx->nccl, ctx->stream));
        }
        BK_NCCL(ctx, ncclGroupEnd());
        return 0;
    }
    // host-staged communicator: own block device-to-device, the others through the proxy thread, in the stream
    BK_HIP(ctx, hipMemcpyAsync(recvbuf + rdispl[me], sendbuf + sdispl[me], scount[me] * sizeof(double),
                               hipMemcpyDeviceToDevice, ctx->stream));
    CommProxy* p = nullptr;
    BK_TRY(proxy_get(ctx, &p));
    ProxyOp o;
    o.kind = 2;
    o.scount.assign(scount, scount + R); o.rcount.assign(rcount, rcount + R);
    o.sdispl.assign(R, 0); o.rdispl.assign(R, 0);
    size_t stot = 0, rtot = 0;
    for (int q = 0; q < R; ++q) {
        if (q == me) continue;
        o.sdispl[q] = stot; stot += scount[q];
        o.rdispl[q] = rtot; rtot += rcount[q];
    }
    BK_TRY(proxy_grow(ctx, ctx->stream, &p->as, &p->ar2, &p->acap, std::max<size_t>(std::max(stot, rtot), 1)));
    for (int q = 0; q < R; ++q)
        if (q != me && scount[q])
            BK_HIP(ctx, hipMemcpyAsync(p->as + o.sdispl[q], sendbuf + sdispl[q], scount[q] * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    BK_TRY(proxy_submit(ctx, p, ctx->stream, o));
    for (int q = 0; q < R; ++q)
        if (q != me && rcount[q])
            BK_HIP(ctx, hipMemcpyAsync(recvbuf + rdispl[q], p->ar2 + o.rdispl[q], rcount[q] * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

static int ctx_init_common(bk_ctx* ctx, int device, void* stream) {
    ctx->device = device;
    BK_HIP(ctx, hipSetDevice(device));
    // NULL = the device's default (null) stream, so that work enqueued by the caller's runtime (PyTorch's default
    // stream, Julia's AMDGPU default queue) is ordered with the library's kernels without explicit events.
    ctx->stream = (hipStream_t)stream;
    ctx->own_stream = false;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->num_cu = cus;
    }
    BK_HIP(ctx, hipMalloc(&ctx->d_partials, sizeof(double) * kPartialDoubles));
    BK_HIP(ctx, hipMalloc(&ctx->d_red, sizeof(double) * kRedSlots));
    BK_HIP(ctx, hipHostMalloc(&ctx->h_red, sizeof(double) * kRedSlots, hipHostMallocMapped));
    {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, ctx->h_red, 0) == hipSuccess) ctx->h_red_dev = static_cast<double*>(dp);
        if (hipHostMalloc(&ctx->h_rec, sizeof(double) * kRecChunks * (kMaxBasis + 2), hipHostMallocMapped) == hipSuccess) {
            void* dr = nullptr;
            if (hipHostGetDevicePointer(&dr, ctx->h_rec, 0) == hipSuccess) ctx->h_rec_dev = static_cast<double*>(dr);
        }
        (void)hipGetLastError();
    }
    return 0;
}

bk_ctx* ctx_lane(bk_ctx* ctx) {
    if (ctx->comm == COMM_HOST && ctx->nranks > 1 && !(ctx->lane_allreduce && ctx->lane_sendrecv)) return nullptr;   // (not an error: one lane)
    if (ctx->lane2) {
        ctx->lane2->opts = ctx->opts;                 // the lane follows the context's options
        ctx->lane2->prof = ctx->prof;
        return ctx->lane2;
    }
    hipStream_t st = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        set_error(ctx, "second lane: stream creation failed");
        return nullptr;
    }
    bk_ctx* l = new bk_ctx();
    // (every failure path below frees the lane, its stream included, through bk_ctx_destroy)
    const int si = ctx_init_common(l, ctx->device, st);
    l->stream = st;
    l->own_stream = true;
    if (si != 0) {
        set_error(ctx, "second lane: %s", l->err.c_str());
        (void)bk_ctx_destroy(l);
        return nullptr;
    }
    l->num_cu = ctx->num_cu;
    l->opts = ctx->opts;
    l->prof = ctx->prof;
    // ranks: the lane gets its OWN communicator -- its collectives are issued by another host thread and must not be
    // matched against the context's.  RCCL: ncclCommSplit of the context's communicator (collective: every rank creates
    // its lane at the same call); host-staged test communicator: the same callbacks with user = lane number, which selects
    // the lane's gloo group on the Python side (hostcomm.py).
    l->rank = ctx->rank;
    l->nranks = ctx->nranks;
    l->comm = ctx->comm;
    if (ctx->comm == COMM_RCCL && ctx->nranks > 1) {
        ncclResult_t r = ncclCommSplit(ctx->nccl, 0, ctx->rank, &l->nccl, nullptr);
        if (r != ncclSuccess) {
            set_error(ctx, "second lane: ncclCommSplit failed: %s", ncclGetErrorString(r));
            l->nccl = nullptr;
            (void)bk_ctx_destroy(l);
            return nullptr;
        }
    } else if (ctx->comm == COMM_HOST) {
        // the lane's callbacks were registered by the client (bk_ctx_set_lane_comm); without them there is no lane
        l->h_allreduce = ctx->lane_allreduce;
        l->h_sendrecv = ctx->lane_sendrecv;
        l->h_user = ctx->lane_user;
    }
    ctx->lane2 = l;
    return l;
}

void ctx_lane_merge(bk_ctx* ctx, bk_ctx* lane) {
    prof_resolve(lane);
    if (ctx->prof)
        for (auto& kv : lane->prof_entries) {
            ProfEntry& e = ctx->prof_entries[kv.first];
            e.ms += kv.second.ms; e.calls += kv.second.calls; e.bytes += kv.second.bytes;
        }
    lane->prof_entries.clear();
    ctx->diag.block_steps += lane->diag.block_steps; ctx->diag.block_truncated += lane->diag.block_truncated;
    ctx->diag.block_unconsumed += lane->diag.block_unconsumed; ctx->diag.check_mismatch += lane->diag.check_mismatch;
    lane->diag = bk_ctx::Diag();
    if (!lane->block_log.empty()) {
        for (size_t i = 0; i < lane->block_log.size(); ++i) {
            double v = lane->block_log[i];
            if (i % bk_ctx::kBlockLogRec == 0) v += ctx->block_log_solves;      // the lane numbered its solves from 1
            ctx->block_log.push_back(v);
        }
        ctx->block_log_solves += lane->block_log_solves;
        lane->block_log.clear();
        lane->block_log_solves = 0;
    }
    if (!lane->hist.empty()) {
        // the lane numbered its solves from 1: renumber them behind the context's own
        for (double v : lane->hist) {
            if (v < 0.0) { ctx->hist_solves += 1; ctx->hist.push_back(-(double)ctx->hist_solves); }
            else ctx->hist.push_back(v);
        }
        lane->hist.clear();
        lane->hist_solves = 0;
    }
}

}  // namespace bk

using namespace bk;

extern "C" {

int bk_version(void) { return 100; }

int bk_ctx_create(bk_ctx** out, int device, void* stream) {
    if (!out) return -1;
    bk_ctx* ctx = new bk_ctx();
    int s = ctx_init_common(ctx, device, stream);
    if (s != 0) {
        fprintf(stderr, "bkhip: bk_ctx_create failed: %s\n", ctx->err.c_str());
        delete ctx;
        *out = nullptr;
        return s;
    }
    *out = ctx;
    return 0;
}

int bk_comm_unique_id(void* id128) {
    static_assert(sizeof(ncclUniqueId) <= BK_UNIQUE_ID_BYTES, "unique id does not fit");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return -1;
    memset(id128, 0, BK_UNIQUE_ID_BYTES);
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int bk_ctx_create_dist(bk_ctx** out, int device, void* stream, int rank, int nranks, const void* id128) {
    if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return -1;
    bk_ctx* ctx = new bk_ctx();
    int s = ctx_init_common(ctx, device, stream);
    if (s == 0) {
        ncclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        ncclResult_t r = ncclCommInitRank(&ctx->nccl, nranks, id, rank);
        if (r != ncclSuccess) s = set_error(ctx, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    if (s != 0) {
        fprintf(stderr, "bkhip: bk_ctx_create_dist failed: %s\n", ctx->err.c_str());
        delete ctx;
        *out = nullptr;
        return s;
    }
    ctx->rank = rank;
    ctx->nranks = nranks;
    ctx->comm = COMM_RCCL;
    *out = ctx;
    return 0;
}

int bk_ctx_create_hostcomm(bk_ctx** out, int device, void* stream, int rank, int nranks, bk_allreduce_fn allreduce,
                           bk_sendrecv_fn sendrecv, void* user) {
    if (!out || !allreduce || !sendrecv || nranks < 1 || rank < 0 || rank >= nranks) return -1;
    bk_ctx* ctx = new bk_ctx();
    int s = ctx_init_common(ctx, device, stream);
    if (s != 0) {
        fprintf(stderr, "bkhip: bk_ctx_create_hostcomm failed: %s\n", ctx->err.c_str());
        delete ctx;
        *out = nullptr;
        return s;
    }
    ctx->rank = rank;
    ctx->nranks = nranks;
    ctx->comm = COMM_HOST;
    ctx->h_allreduce = allreduce;
    ctx->h_sendrecv = sendrecv;
    ctx->h_user = user;
    *out = ctx;
    return 0;
}

int bk_comm_info(bk_ctx* ctx, int* kind, int* rank, int* nranks) {
    if (!ctx) return -1;
    int r = ctx->rank, n = ctx->nranks;
    if (ctx->comm == COMM_RCCL && ctx->nccl) {
        BK_NCCL(ctx, ncclCommUserRank(ctx->nccl, &r));
        BK_NCCL(ctx, ncclCommCount(ctx->nccl, &n));
    }
    if (kind) *kind = (int)ctx->comm;
    if (rank) *rank = r;
    if (nranks) *nranks = n;
    return 0;
}

int bk_comm_probe(bk_ctx* ctx, int what, size_t count, int reps, double* us_per_call) {
    if (!ctx || !us_per_call || reps < 1 || count < 1) return -1;
    *us_per_call = 0.0;
    if (ctx->comm == COMM_NONE || ctx->nranks == 1) return 0;
    WsGuard ws(ctx);
    double *v = nullptr, *lo = nullptr, *hi = nullptr;
    if (what == 0) {
        if (count > (size_t)kRedSlots) return set_error(ctx, "bk_comm_probe: at most %d doubles per all-reduce", kRedSlots);
        BK_HIP(ctx, hipMemsetAsync(ctx->d_red, 0, count * sizeof(double), ctx->stream));
    } else if (what == 1) {
        BK_TRY(ws.get(2 * count, &v));
        BK_TRY(ws.get(count, &lo));
        BK_TRY(ws.get(count, &hi));
        BK_HIP(ctx, hipMemsetAsync(v, 0, 2 * count * sizeof(double), ctx->stream));
    } else {
        return set_error(ctx, "bk_comm_probe: unknown probe %d", what);
    }
    std::vector<double> hbuf(what == 0 ? count : 0, 0.0);
    for (int pass = 0; pass < 2; ++pass) {           // pass 0 = warm-up (connection set-up of the first call)
        const int n = pass == 0 ? 2 : reps;
        BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {
            if (what == 0) {
                BK_TRY(comm_allreduce_dev(ctx, ctx->stream, ctx->d_red, (int)count, 0));
            } else {
                BK_TRY(halo_exchange(ctx, ctx->stream, v, count, 2, 1, lo, hi));
            }
        }
        BK_TRY(ctx_sync(ctx));
        if (pass == 1)
            *us_per_call = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    }
    return 0;
}

int bk_ctx_destroy(bk_ctx* ctx) {
    if (!ctx) return 0;
    if (ctx->lane2) { bk_ctx* l = ctx->lane2; ctx->lane2 = nullptr; (void)bk_ctx_destroy(l); }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    proxy_destroy(ctx);
    for (auto& kv : ctx->pool_all) (void)hipFree(kv.first);
    for (auto& kv : ctx->prof_entries)
        for (auto& pr : kv.second.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    if (ctx->d_red) (void)hipFree(ctx->d_red);
    if (ctx->h_red) (void)hipHostFree(ctx->h_red);
    if (ctx->h_rec) (void)hipHostFree(ctx->h_rec);
    blas_release(ctx);
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    if (ctx->ev_ready) (void)hipEventDestroy(ctx->ev_ready);
    if (ctx->ev_halo) (void)hipEventDestroy(ctx->ev_halo);
    if (ctx->nccl) (void)ncclCommDestroy(ctx->nccl);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

const char* bk_last_error(bk_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int bk_ctx_sync(bk_ctx* ctx) {
    if (!ctx) return -1;
    return ctx_sync(ctx);
}

int bk_ctx_set_lane_comm(bk_ctx* ctx, bk_allreduce_fn allreduce, bk_sendrecv_fn sendrecv, void* user) {
    if (!ctx) return -1;
    if (ctx->comm != COMM_HOST) return set_error(ctx, "bk_ctx_set_lane_comm: host-staged contexts only (RCCL lanes split the communicator)");
    if (ctx->lane2) return set_error(ctx, "bk_ctx_set_lane_comm: the second lane already exists");
    ctx->lane_allreduce = allreduce;
    ctx->lane_sendrecv = sendrecv;
    ctx->lane_user = user;
    return 0;
}

int bk_ctx_set_option(bk_ctx* ctx, const char* key, double value) {
    if (!ctx || !key) return -1;
    if (double* d = ctx->diag_slot(key)) { *d = value; return 0; }
    ctx->opts[key] = value;
    ctx->lanes_warm.clear();                 // (another option set may ask the pools for other buffers: linsolve2 warms them again)
    return 0;
}

int bk_ctx_get_option(bk_ctx* ctx, const char* key, double* value) {
    if (!ctx || !key || !value) return -1;
    if (const double* d = ctx->diag_slot(key)) { *value = *d; return 0; }
    auto it = ctx->opts.find(key);
    if (it == ctx->opts.end()) return set_error(ctx, "unknown option %s", key);
    *value = it->second;
    return 0;
}

int bk_prof_enable(bk_ctx* ctx, int on) {
    if (!ctx) return -1;
    if (!on) prof_resolve(ctx);
    ctx->prof = on != 0;
    return 0;
}

int bk_prof_reset(bk_ctx* ctx) {
    if (!ctx) return -1;
    prof_resolve(ctx);
    ctx->prof_entries.clear();
    return 0;
}

int bk_solver_history(bk_ctx* ctx, double* buf, size_t cap, size_t* n, int reset) {
    if (!ctx) return -1;
    if (n) *n = ctx->hist.size();
    if (buf)
        for (size_t i = 0; i < ctx->hist.size() && i < cap; ++i) buf[i] = ctx->hist[i];
    if (reset) { ctx->hist.clear(); ctx->hist_solves = 0; }
    return 0;
}

int bk_solver_block_log(bk_ctx* ctx, double* buf, size_t cap, size_t* n, int reset) {
    if (!ctx) return -1;
    if (n) *n = ctx->block_log.size();
    if (buf)
        for (size_t i = 0; i < ctx->block_log.size() && i < cap; ++i) buf[i] = ctx->block_log[i];
    if (reset) { ctx->block_log.clear(); ctx->block_log_solves = 0; }
    return 0;
}

int bk_prof_get(bk_ctx* ctx, const char* name, double* total_ms, long long* calls, double* alg_bytes) {
    if (!ctx || !name) return -1;
    prof_resolve(ctx);
    auto it = ctx->prof_entries.find(name);
    if (it == ctx->prof_entries.end()) {
        if (total_ms) *total_ms = 0.0;
        if (calls) *calls = 0;
        if (alg_bytes) *alg_bytes = 0.0;
        return 0;
    }
    if (total_ms) *total_ms = it->second.ms;
    if (calls) *calls = it->second.calls;
    if (alg_bytes) *alg_bytes = it->second.bytes;
    return 0;
}

int bk_malloc(bk_ctx* ctx, size_t n, double** out) {
    if (!ctx || !out) return -1;
    BK_HIP(ctx, hipSetDevice(ctx->device));
    BK_HIP(ctx, hipMalloc(out, (n ? n : 1) * sizeof(double)));
    return 0;
}

int bk_free(bk_ctx* ctx, double* p) {
    if (!ctx) return -1;
    if (p) BK_HIP(ctx, hipFree(p));
    return 0;
}

int bk_upload(bk_ctx* ctx, double* dst, const double* src, size_t n) {
    BK_HIP(ctx, hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int bk_download(bk_ctx* ctx, double* dst, const double* src, size_t n) {
    BK_HIP(ctx, hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
