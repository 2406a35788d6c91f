// DirectQueue: the sampler's own user-mode AQL queue (host side only).
//
// Why: the steady state of a run is two DEPENDENT update kernels of 5-6 us per generation.  A HIP launch call costs the host
// 2.4-4.8 us (and now and then 10-30 us), so a call that starts from a drained queue -- the 20 generations the driver times --
// runs host-paced.  Here the library writes the 64-byte AQL dispatch packets itself: kernel
// arguments into a ring in device memory (through the PCIe BAR), the packet into the queue's ring, one doorbell per
// generation.  ~0.5 us of host work per dispatch (half of it the read-back that proves the arguments have landed), nothing of the HIP runtime on the path (tools/micro/aql_direct.cpp measures
// the pieces: a dependent empty dispatch costs 1.92 us this way against 2.62 us through hipLaunchKernelGGL; kernel arguments
// in HOST memory cost 27 us per 4096-wavefront dispatch, hence the device ring).
//
// What it is not: a second code path for the kernels.  The kernel objects are the ones HIP loaded from this library's own fat
// binary (found through the HSA loader's executable list), every packet carries the barrier bit, and memory is HIP's.  Fences are the
// caller's choice per packet: agent-scope acquire + release like a HIP stream's, or the acquire only -- the sampler's steady state,
// whose kernels send what their successor reads through agent-scope stores (sampler.hip: g_dq_update_fence).  The queue keeps track of
// release-less packets: drain() puts a fenced empty kernel behind them unless a later kernel on every XCD has released since.
// Ordering against the sampler's HIP stream is by the host: the sampler drains one before it uses the other (transitions happen at
// the end of burn-in and at the API boundary only).  BPM_QUEUE_INFLIGHT=n bounds the dispatches between two drains (for tools that
// sit between this queue and the hardware queue: rocprofv3's counter collection, see inflight_cap(); chosen automatically, 64, when
// ROCPROF_COUNTER_COLLECTION is set in the environment).
#pragma once
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_ven_amd_loader.h>
#include <x86intrin.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace bpm {

struct DqKernel {
    uint64_t object = 0;
    uint32_t kernarg_size = 0, group = 0, priv = 0;
    std::string name;
};

class DirectQueue {
  public:
#ifndef BPM_QUEUE_PACKETS
#define BPM_QUEUE_PACKETS 1024          // (experiment builds change it: the rocprofv3 --pmc threshold follows it, see inflight_cap())
#endif
    static constexpr uint32_t QUEUE_PACKETS = BPM_QUEUE_PACKETS, SLOT_BYTES = 3072, N_SLOTS = QUEUE_PACKETS;
    // SYSTEM: the fences named by ACQUIRE / RELEASE are system scope instead of agent scope (push exchange: rows written into other
    // ranks' memory must have been performed before the completion is announced, and rows peers wrote here must be seen)
    enum : int { ACQUIRE = 1, RELEASE = 2, FENCED = 3, SYSTEM = 4 };

    // one queue per HIP device of the process; nullptr when the queue cannot be had (reason in why())
    static DirectQueue* for_device(int hip_dev) {
        static std::mutex mu;
        static std::map<int, DirectQueue*> all;
        std::lock_guard<std::mutex> lk(mu);
        auto it = all.find(hip_dev);
        if (it != all.end()) return it->second;
        DirectQueue* q = new DirectQueue();
        if (!q->init(hip_dev)) {
            static const bool verbose = getenv("BPM_VERBOSE") != nullptr;
            if (verbose) fprintf(stderr, "[bipymc_amd] direct AQL queue not available on device %d (%s): HIP stream launches\n", hip_dev, q->why_.c_str());
            last_reason() = q->why_;
            delete q;
            q = nullptr;
        }
        all[hip_dev] = q;
        return q;
    }
    static std::string& last_reason() { static std::string r; return r; }
    // A queue of its own for ONE sampler (the ranks of a local test group: R handles of one process whose kernels must be able to wait
    // for each other across queues, like the ranks of a multi-GPU world do).  The caller owns it: destroy_private().
    static DirectQueue* create_private(int hip_dev) {
        DirectQueue* q = new DirectQueue();
        if (!q->init(hip_dev)) { last_reason() = q->why_; delete q; return nullptr; }
        q->private_ = true;
        return q;
    }
    static void destroy_private(DirectQueue* q) {
        if (!q || !q->private_) return;
        if (q->q_) (void)hsa_queue_destroy(q->q_);
        for (hsa_signal_t sg : {q->done_, q->tsig_[0], q->tsig_[1]}) if (sg.handle) (void)hsa_signal_destroy(sg);
        for (auto& es : q->epoch_sig_) if (es.handle) (void)hsa_signal_destroy(es);
        if (q->kernarg_) (void)hsa_amd_memory_pool_free(q->kernarg_);
        delete q;
    }

    // the kernel HIP would launch for this host function, as the dispatch packet names it; nullptr if it cannot be located
    const DqKernel* kernel(const void* host_fn) {
        std::lock_guard<std::recursive_mutex> lk(mu_);
        auto it = kernels_.find(host_fn);
        if (it != kernels_.end()) return it->second.object ? &it->second : nullptr;
        DqKernel k;
        hipFunction_t fn = nullptr;
        if (hipGetFuncBySymbol(&fn, host_fn) != hipSuccess) { (void)hipGetLastError(); }      // (makes HIP load the code object)
        const char* nm = hipKernelNameRefByPtr(host_fn, nullptr);
        if (fn && nm) {
            k.name = nm;
            Lookup lk{this, k.name + ".kd", &k, false};
            (void)loader_.hsa_ven_amd_loader_iterate_executables(&DirectQueue::exe_cb, &lk);
            if (!lk.found) k.object = 0;
        }
        auto& slot = kernels_[host_fn];
        slot = k;
        return slot.object ? &slot : nullptr;
    }

    // ... and a kernel of a module loaded at run time (hipModuleLoadData: a caller's likelihood compiled by hiprtc, user_likelihood.h), by its lowered name;
    // the name must be unique among the process's code objects (the module's device code lives in a namespace of its own)
    const DqKernel* kernel_by_name(const std::string& lowered) {
        std::lock_guard<std::recursive_mutex> lk(mu_);
        auto it = named_.find(lowered);
        if (it != named_.end()) return it->second.object ? &it->second : nullptr;
        DqKernel k;
        k.name = lowered;
        Lookup lkp{this, lowered + ".kd", &k, false};
        (void)loader_.hsa_ven_amd_loader_iterate_executables(&DirectQueue::exe_cb, &lkp);
        if (!lkp.found) k.object = 0;
        auto& slot = named_[lowered];
        slot = k;
        return slot.object ? &slot : nullptr;
    }
    void forget_named(const std::string& lowered) { std::lock_guard<std::recursive_mutex> lk(mu_); named_.erase(lowered); }      // (its module is being unloaded)

    // One dispatch.  `args` = the explicit kernel arguments laid out as the kernel's kernarg segment has them (natural
    // alignment), `nbytes` their size; the hidden arguments a kernel that asks for blockDim / gridDim reads are appended here.
    // fence: ACQUIRE | RELEASE at agent scope (what a HIP stream puts around every kernel).  sig: 0 / 1 = this dispatch
    // carries timing signal 0 / 1 (see dispatch_end_ns), -1 = none.  The doorbell is rung by flush().
    int launch(const DqKernel& k, uint32_t grid_x, uint32_t grid_y, uint32_t block, const void* args, size_t nbytes, int fence, int sig = -1) {
        std::lock_guard<std::recursive_mutex> lk(mu_);          // (samplers of several threads may share a device's queue; uncontended: ~20 ns)
        if (failed_) return -1;
        const size_t hidden_at = (nbytes + 7) & ~size_t(7);
        const bool hidden = k.kernarg_size >= hidden_at + 66;
        if (nbytes > k.kernarg_size) nbytes = k.kernarg_size;       // (a kernel that ignores trailing arguments has a shorter segment)
        if (k.kernarg_size > SLOT_BYTES) { why_ = "kernel argument block of " + k.name + " does not fit"; failed_ = true; return -1; }
        if (!next_slot()) return -1;
        // a release at the end of a kernel that ran on every XCD (>= 64 workgroups) writes back what EARLIER release-less kernels left
        // in the L2s as well (every packet waits for its predecessor: barrier bit)
        if (!(fence & RELEASE)) unreleased_ = true;
        else if ((uint64_t)grid_x * grid_y >= 64) unreleased_ = false;
        char* slot = kernarg_ + (size_t)(widx_ % N_SLOTS) * SLOT_BYTES;
        std::memcpy(slot, args, nbytes);
        size_t written = nbytes;
        if (hidden) {
            struct Hidden {
                uint32_t block_count[3];
                uint16_t group_size[3], remainder[3];
                uint8_t reserved[16];
                uint64_t global_offset[3];
                uint16_t grid_dims;
            } h;
            static_assert(offsetof(Hidden, global_offset) == 40 && offsetof(Hidden, grid_dims) == 64, "implicit kernel argument layout (code object v5)");
            std::memset(&h, 0, sizeof(h));
            h.block_count[0] = grid_x; h.block_count[1] = grid_y; h.block_count[2] = 1;
            h.group_size[0] = (uint16_t)block; h.group_size[1] = 1; h.group_size[2] = 1;
            h.grid_dims = grid_y > 1 ? 2 : 1;
            std::memcpy(slot + hidden_at, &h, 66);
            written = hidden_at + 66;
        }
        last_written_ = reinterpret_cast<volatile uint32_t*>(slot + ((written - 1) & ~size_t(3)));
        auto* p = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(q_->base_address) + (widx_ & (q_->size - 1));
        p->workgroup_size_x = (uint16_t)block; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
        p->grid_size_x = grid_x * block; p->grid_size_y = grid_y; p->grid_size_z = 1;
        p->private_segment_size = k.priv; p->group_segment_size = k.group;
        p->kernel_object = k.object; p->kernarg_address = slot; p->reserved2 = 0;
        hsa_signal_t none{0};
        if (sig == 0 || sig == 1) { hsa_signal_store_relaxed(tsig_[sig], 1); tsig_armed_[sig] = true; }
        p->completion_signal = (sig == 0 || sig == 1) ? tsig_[sig] : none;
        const int scope = (fence & SYSTEM) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
        const uint16_t hdr = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                        (((fence & ACQUIRE) ? scope : HSA_FENCE_SCOPE_NONE) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                        (((fence & RELEASE) ? scope : HSA_FENCE_SCOPE_NONE) << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        pending_header_[n_unpublished_] = (uint32_t)hdr | ((uint32_t)(3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS) << 16);
        pending_packet_[n_unpublished_] = reinterpret_cast<uint32_t*>(p);
        // (the packet's first word still says INVALID: the packet processor stops in front of it until flush() publishes it)
        ++n_unpublished_;
        ++widx_;
        busy_ = true;
        // one doorbell never covers packets on both sides of the ring's end (a queue-intercepting tool -- rocprofv3 -- copies the
        // packets of a doorbell as one linear range)
        if (n_unpublished_ >= batch_cap() || (widx_ & (q_->size - 1)) == 0) flush();
        // BPM_QUEUE_INFLIGHT=n: never more than n dispatches between two drains (a throttle for tools that sit between this queue and
        // the hardware queue -- rocprofv3's counter collection: tools/profile_bench.sh)
        if (inflight_cap() && !in_drain_ && ++since_drain_ >= inflight_cap()) { since_drain_ = 0; if (drain() != 0) return -1; busy_ = true; }
        return 0;
    }

    // publish what launch() has written: arguments first (store fence + read-back through the BAR: the posted writes have
    // reached device memory), then the packet headers, then the doorbell
    void flush() {
        std::lock_guard<std::recursive_mutex> lk(mu_);
        if (n_unpublished_ == 0) return;
        _mm_sfence();
        if (last_written_) { const uint32_t sink = *last_written_; (void)sink; }
        for (uint32_t i = 0; i < n_unpublished_; ++i) __atomic_store_n(pending_packet_[i], pending_header_[i], __ATOMIC_RELEASE);
        n_unpublished_ = 0;
        hsa_queue_store_write_index_screlease(q_, widx_);
        hsa_signal_store_screlease(q_->doorbell_signal, (hsa_signal_value_t)(widx_ - 1));
    }

    // everything dispatched so far has finished and is visible to the host and to HIP streams (system-scope release)
    int drain(double timeout_s = -1.0) {
        std::lock_guard<std::recursive_mutex> lk(mu_);
        if (timeout_s < 0.0) timeout_s = wait_limit_s();
        if (!busy_) return 0;
        if (failed_) return -1;
        struct InDrain { bool& f; bool old; InDrain(bool& x) : f(x), old(x) { f = true; } ~InDrain() { f = old; } } guard(in_drain_);
        since_drain_ = 0;
        if (!next_slot()) return -1;
        if (unreleased_) {
            // packets without a release fence were dispatched: an (empty) kernel with acquire + release first, so that the caches end up as
            // after the last kernel of a HIP stream (history rows of non-temporal stores may still wait in an L2)
            if (!fence_kernel_.object) { why_ = "release-less packets without a fence kernel"; failed_ = true; return -1; }
            uint64_t zero = 0;
            if (launch(fence_kernel_, 64, 1, 64, &zero, sizeof(zero), FENCED) != 0) return -1;
            unreleased_ = false;
            if (!next_slot()) return -1;
        }
        // The barrier packet takes the same road as every other packet: its header joins the pending list behind whatever next_slot() left
        // there (an epoch marker), and flush() publishes the list IN ORDER, write index and doorbell last.  Round 3 wrote this one header
        // directly and once rang the doorbell over an unpublished marker -- the packet processor then waits at an INVALID header for good
        // (1 in 256 drains of that kind; tests/test_gpu_api.py::test_drains_at_every_position_of_the_queues_epochs).  No packet header is
        // written outside publish order anywhere in this file now.
        auto* b = reinterpret_cast<hsa_barrier_and_packet_t*>(q_->base_address) + (widx_ & (q_->size - 1));
        std::memset(reinterpret_cast<char*>(b) + 4, 0, sizeof(*b) - 4);
        hsa_signal_store_relaxed(done_, 1);
        b->completion_signal = done_;
        const uint16_t hdr = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                        (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                        (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        if (n_unpublished_ >= MAX_UNPUBLISHED) flush();
        pending_header_[n_unpublished_] = (uint32_t)hdr;
        pending_packet_[n_unpublished_] = reinterpret_cast<uint32_t*>(b);
        ++n_unpublished_;
        ++widx_;
        flush();
        const auto t0 = std::chrono::steady_clock::now();
        while (hsa_signal_load_scacquire(done_) > 0) {                       // polling: a parked thread wakes up on a slow clock
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
                why_ = "timeout waiting for the AQL queue to drain (a tool sitting between this queue and the hardware? BPM_QUEUE_INFLIGHT=64 bounds the "
                       "dispatches in flight, BPM_DIRECT_QUEUE=0 launches on the HIP stream)";
                failed_ = true; return -1;
            }
        }
        busy_ = false;
        return 0;
    }
    // the empty kernel drain() runs behind release-less packets (set once by the owner of the code object); without it launch() callers
    // must keep the release fence
    bool set_fence_kernel(const void* host_fn) { if (const DqKernel* k = kernel(host_fn)) fence_kernel_ = *k; return fence_kernel_.object != 0; }
    bool has_fence_kernel() const { return fence_kernel_.object != 0; }
    bool busy() const { return busy_; }
    bool failed() const { return failed_; }
    // After a failure (a wait ran into its limit): make sure NOTHING dispatched on this queue can still touch memory, so that its users may
    // free their buffers.  A kernel that is slow or stalled behind a tool, not dead, is invisible to hipStreamSynchronize / hipFree: it would
    // go on writing freed memory -- a GPU memory fault on exactly the path that was already in trouble.  hsa_queue_inactivate aborts what
    // is pending and stops the packet processor from taking more; the queue stays dead for the rest of the process (samplers launch on
    // their HIP streams from then on).  -> true: quiet (or never failed and drained), false: could not be quiesced -- the caller must
    // LEAK what the queue's kernels may reference.
    bool quiesce() {
        std::lock_guard<std::recursive_mutex> lk(mu_);
        if (!failed_) return !busy_ || drain() == 0 || quiesce();
        if (dead_) return true;
        if (test_refuse_quiesce_) return false;
        if (q_ && hsa_queue_inactivate(q_) == HSA_STATUS_SUCCESS) { dead_ = true; busy_ = false; n_unpublished_ = 0; return true; }
        return false;
    }
    bool dead() const { return dead_; }
#ifdef BPM_TEST_HOOKS
    // test hook (bpm_debug_queue_pad): no-op barrier packets until the next packet would take position `pos` of its epoch of 256 (pos < 255);
    // -> the write index.  Lets a test put a drain's packets at a chosen place of the ring.
    int64_t test_pad_to(uint32_t pos) {
        std::lock_guard<std::recursive_mutex> lk(mu_);
        if (failed_ || pos >= EPOCH - 1) return -1;
        for (;;) {
            if (!next_slot()) return -1;
            if (widx_ % EPOCH == pos) break;
            auto* b = reinterpret_cast<hsa_barrier_and_packet_t*>(q_->base_address) + (widx_ & (q_->size - 1));
            std::memset(reinterpret_cast<char*>(b) + 4, 0, sizeof(*b) - 4);
            pending_header_[n_unpublished_] = (uint32_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER));
            pending_packet_[n_unpublished_] = reinterpret_cast<uint32_t*>(b);
            ++n_unpublished_;
            ++widx_;
            busy_ = true;
            if (n_unpublished_ >= MAX_UNPUBLISHED - 1 || (widx_ & (q_->size - 1)) == 0) flush();
        }
        flush();
        return (int64_t)widx_;
    }
    // test hook (bpm_debug_fail_queue): behave as if a drain had timed out; refuse != 0: and as if the queue could not be inactivated
    void test_mark_failed(bool refuse) { std::lock_guard<std::recursive_mutex> lk(mu_); failed_ = true; why_ = "failure injected by the test hook"; test_refuse_quiesce_ = refuse; }
#endif
    const std::string& why() const { return why_; }

    // end-of-kernel time stamps (ns, one clock) of the dispatches that carried timing signal 0 and 1; call after drain()
    bool dispatch_end_ns(double* t0_ns, double* t1_ns) {
        std::lock_guard<std::recursive_mutex> lk(mu_);
        if (!tsig_armed_[0] || !tsig_armed_[1]) return false;
        hsa_amd_profiling_dispatch_time_t a{}, b{};
        if (hsa_amd_profiling_get_dispatch_time(agent_, tsig_[0], &a) != HSA_STATUS_SUCCESS) return false;
        if (hsa_amd_profiling_get_dispatch_time(agent_, tsig_[1], &b) != HSA_STATUS_SUCCESS) return false;
        *t0_ns = (double)a.end * tick_ns_;
        *t1_ns = (double)b.end * tick_ns_;
        return true;
    }
    void disarm_timing() { tsig_armed_[0] = tsig_armed_[1] = false; }

  private:
    static constexpr uint32_t MAX_UNPUBLISHED = 8;
    static constexpr uint32_t EPOCH = 256, N_EPOCH = QUEUE_PACKETS / EPOCH;
    struct Lookup { DirectQueue* self; std::string sym; DqKernel* out; bool found; };

    static hsa_status_t exe_cb(hsa_executable_t exe, void* data) {
        Lookup* lk = static_cast<Lookup*>(data);
        hsa_executable_symbol_t sym;
        if (hsa_executable_get_symbol_by_name(exe, lk->sym.c_str(), &lk->self->agent_, &sym) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
        DqKernel& k = *lk->out;
        if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object) != HSA_STATUS_SUCCESS || !k.object) return HSA_STATUS_SUCCESS;
        hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_size);
        hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group);
        hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv);
        lk->found = true;
        return HSA_STATUS_INFO_BREAK;
    }
    struct AgentPick { uint32_t bdf; uint32_t domain; hsa_agent_t gpu, cpu; bool have_gpu, have_cpu; };
    static hsa_status_t agent_cb(hsa_agent_t a, void* data) {
        AgentPick* p = static_cast<AgentPick*>(data);
        hsa_device_type_t t;
        if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
        if (t == HSA_DEVICE_TYPE_CPU && !p->have_cpu) { p->cpu = a; p->have_cpu = true; }
        if (t == HSA_DEVICE_TYPE_GPU) {
            uint32_t bdf = 0, dom = 0;
            hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
            hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
            if (!p->have_gpu && bdf == p->bdf && dom == p->domain) { p->gpu = a; p->have_gpu = true; }
        }
        return HSA_STATUS_SUCCESS;
    }
    static hsa_status_t pool_cb(hsa_amd_memory_pool_t pool, void* data) {
        hsa_amd_segment_t seg;
        if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
        uint32_t fl = 0; bool alloc = false;
        hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
        hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
        hsa_amd_memory_pool_t* out = static_cast<hsa_amd_memory_pool_t*>(data);
        if (alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !out->handle) *out = pool;
        return HSA_STATUS_SUCCESS;
    }

    bool init(int hip_dev) {
        if (const char* e = getenv("BPM_DIRECT_QUEUE")) if (atoi(e) == 0) { why_ = "BPM_DIRECT_QUEUE=0"; return false; }
        int bus = 0, dev = 0, dom = 0;
        if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, hip_dev) != hipSuccess || hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, hip_dev) != hipSuccess ||
            hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, hip_dev) != hipSuccess) { (void)hipGetLastError(); why_ = "no PCI address of the HIP device"; return false; }
        if (hsa_init() != HSA_STATUS_SUCCESS) { why_ = "hsa_init failed"; return false; }           // (reference counted: HIP holds the runtime open already)
        AgentPick pick{};
        pick.bdf = ((uint32_t)bus << 8) | ((uint32_t)dev << 3); pick.domain = (uint32_t)dom;
        if (hsa_iterate_agents(&DirectQueue::agent_cb, &pick) != HSA_STATUS_SUCCESS || !pick.have_gpu || !pick.have_cpu) { why_ = "HSA agent of the HIP device not found"; return false; }
        agent_ = pick.gpu; cpu_ = pick.cpu;
        if (hsa_system_get_major_extension_table(HSA_EXTENSION_AMD_LOADER, 1, sizeof(loader_), &loader_) != HSA_STATUS_SUCCESS ||
            !loader_.hsa_ven_amd_loader_iterate_executables) { why_ = "HSA loader extension 1.03 missing"; return false; }
        hsa_amd_memory_pool_t pool{};
        hsa_amd_agent_iterate_memory_pools(agent_, &DirectQueue::pool_cb, &pool);
        if (!pool.handle) { why_ = "no device memory pool"; return false; }
        hsa_amd_memory_pool_access_t acc = HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
        hsa_amd_agent_memory_pool_get_info(cpu_, pool, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &acc);
        if (acc == HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED) { why_ = "device memory is not host-writable (no large BAR)"; return false; }
        if (hsa_amd_memory_pool_allocate(pool, (size_t)N_SLOTS * SLOT_BYTES, 0, reinterpret_cast<void**>(&kernarg_)) != HSA_STATUS_SUCCESS) { why_ = "kernarg ring allocation failed"; return false; }
        if (hsa_amd_agents_allow_access(1, &cpu_, nullptr, kernarg_) != HSA_STATUS_SUCCESS) { why_ = "host access to the kernarg ring refused"; return false; }
        if (hsa_queue_create(agent_, QUEUE_PACKETS, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q_) != HSA_STATUS_SUCCESS) { why_ = "hsa_queue_create failed"; return false; }
        (void)hsa_amd_profiling_set_profiler_enabled(q_, 1);
        if (hsa_signal_create(0, 0, nullptr, &done_) != HSA_STATUS_SUCCESS || hsa_signal_create(0, 0, nullptr, &tsig_[0]) != HSA_STATUS_SUCCESS ||
            hsa_signal_create(0, 0, nullptr, &tsig_[1]) != HSA_STATUS_SUCCESS) { why_ = "hsa_signal_create failed"; return false; }
        for (auto& es : epoch_sig_) if (hsa_signal_create(0, 0, nullptr, &es) != HSA_STATUS_SUCCESS) { why_ = "hsa_signal_create failed"; return false; }
        uint64_t freq = 0;
        hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq);
        tick_ns_ = freq ? 1e9 / (double)freq : 10.0;
        widx_ = hsa_queue_load_write_index_relaxed(q_);
        return true;
    }
    // widx_ names a free packet slot whose kernarg slot is free too.  Packet ring: the processor has taken the packet that used
    // the slot before (read index).  Kernarg ring: the KERNEL that used the slot before has finished -- the read index does not
    // say that (a packet is off the ring when it is launched, and under a queue-intercepting profiler when it is forwarded), so the
    // last slot of every EPOCH of 256 packets is a barrier packet of the queue's own with a completion signal, and an epoch's
    // slots are reused only after the marker that closed their previous use has completed.
    bool next_slot() {
        for (;;) {
            if (widx_ - hsa_queue_load_read_index_scacquire(q_) >= (uint64_t)QUEUE_PACKETS - 8) {
                flush();
                const auto t0 = std::chrono::steady_clock::now();
                while (widx_ - hsa_queue_load_read_index_scacquire(q_) >= (uint64_t)QUEUE_PACKETS - 8)
                    if (timed_out(t0, "timeout waiting for room in the AQL queue")) return false;
            }
            const uint32_t pos = (uint32_t)(widx_ % EPOCH), e = (uint32_t)((widx_ / EPOCH) % N_EPOCH);
            if (pos == 0 && epoch_armed_[e]) {
                flush();
                const auto t0 = std::chrono::steady_clock::now();
                while (hsa_signal_load_scacquire(epoch_sig_[e]) > 0)
                    if (timed_out(t0, "timeout waiting for an epoch of the AQL queue to finish")) return false;
                epoch_armed_[e] = false;
            }
            if (pos != EPOCH - 1) return true;
            auto* b = reinterpret_cast<hsa_barrier_and_packet_t*>(q_->base_address) + (widx_ & (q_->size - 1));
            std::memset(reinterpret_cast<char*>(b) + 4, 0, sizeof(*b) - 4);
            hsa_signal_store_relaxed(epoch_sig_[e], 1);
            epoch_armed_[e] = true;
            b->completion_signal = epoch_sig_[e];
            pending_header_[n_unpublished_] = (uint32_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER));
            pending_packet_[n_unpublished_] = reinterpret_cast<uint32_t*>(b);
            ++n_unpublished_;
            ++widx_;
            if (n_unpublished_ >= MAX_UNPUBLISHED - 1 || (widx_ & (q_->size - 1)) == 0) flush();
        }
    }
    // Packets per doorbell at most: the sampler rings once per generation (2-4 packets).  (Ringing per packet measured the same without a
    // profiler -- 7.31e8 / 7.79e8 at cfg2 either way -- but makes a kernel trace slower: every doorbell is a call into rocprofv3's
    // intercepting queue, 5.44 instead of 5.01 us average kernel duration in the trace of the driver's invocation.)
    static constexpr uint32_t batch_cap() { return MAX_UNPUBLISHED; }
    static uint32_t inflight_cap() {
        // under rocprofv3's counter collection (it exports ROCPROF_COUNTER_COLLECTION=1 to the profiled process) 64 unless told otherwise.
        // Why (profiles/r03_pmc_queue_stall.txt): with a TCC-derived counter (FETCH_SIZE, WRITE_SIZE) the profiler stops completing
        // dispatches once a queue is several hundred profiled dispatches ahead of it -- no progress in 170 s; with SQ counters nothing
        // stalls however far ahead the queue runs, with a 1024- or a 4096-packet ring: the threshold does NOT follow QUEUE_PACKETS / EPOCH,
        // it is the profiler's handling of that counter set.  A HIP stream never gets that far ahead (its launch calls block).
        static const uint32_t v = getenv("BPM_QUEUE_INFLIGHT") ? (uint32_t)std::max(0, atoi(getenv("BPM_QUEUE_INFLIGHT")))
                                  : (getenv("ROCPROF_COUNTER_COLLECTION") && atoi(getenv("ROCPROF_COUNTER_COLLECTION")) != 0 ? 64u : 0u);
        return v;
    }
    static double wait_limit_s() {
        static const double v = getenv("BPM_QUEUE_TIMEOUT_S") ? atof(getenv("BPM_QUEUE_TIMEOUT_S")) : 120.0;
        return v;
    }
    bool timed_out(std::chrono::steady_clock::time_point t0, const char* what) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < wait_limit_s()) return false;
        why_ = what; failed_ = true;
        return true;
    }

    hsa_agent_t agent_{}, cpu_{};
    hsa_ven_amd_loader_1_03_pfn_t loader_{};
    hsa_queue_t* q_ = nullptr;
    char* kernarg_ = nullptr;
    uint64_t widx_ = 0;
    hsa_signal_t done_{}, tsig_[2]{}, epoch_sig_[N_EPOCH]{};
    bool epoch_armed_[N_EPOCH] = {};
    bool tsig_armed_[2] = {false, false};
    double tick_ns_ = 10.0;
    bool busy_ = false, failed_ = false, dead_ = false, test_refuse_quiesce_ = false, private_ = false;
    uint32_t n_unpublished_ = 0;
    uint32_t pending_header_[MAX_UNPUBLISHED]{};
    uint32_t* pending_packet_[MAX_UNPUBLISHED]{};
    volatile uint32_t* last_written_ = nullptr;
    std::string why_;
    std::recursive_mutex mu_;
    std::map<const void*, DqKernel> kernels_;
    std::map<std::string, DqKernel> named_;      // kernels of run-time modules, by lowered name (kernel_by_name)
    DqKernel fence_kernel_;
    bool unreleased_ = false;
    bool in_drain_ = false;
    uint32_t since_drain_ = 0;
};

}  // namespace bpm
