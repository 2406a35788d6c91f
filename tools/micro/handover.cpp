// Microbenchmark (diagnostic): what the hand-over between two "ranks" costs when each rank is an AQL queue of its own on ONE GPU.
//   A. reference: each queue runs dependent work kernels with no hand-over at all
//   B. the push exchange's form: work kernel, then a one-wavefront kernel that announces in the peer's flag and polls its own
//   C. the command processor waits: the work kernel's packet carries a completion signal (decremented when the kernel has finished), the peer's
//      queue holds an AMD barrier-value packet on that signal (HSA_AMD_PACKET_TYPE_BARRIER_VALUE: wait until signal < value) -- no extra dispatch
// DESIGN.md section 10 item 3.  Every wait is bounded.
//   hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 -o build_variants/handover_kernels.hsaco tools/micro/handover_kernels.hip
//   g++ -O2 -std=c++17 -I/opt/rocm/include -o build_variants/handover tools/micro/handover.cpp -L/opt/rocm/lib -lhsa-runtime64 -Wl,-rpath,/opt/rocm/lib
//   ./build_variants/handover build_variants/handover_kernels.hsaco
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#define HCK(e) do { hsa_status_t _s = (e); if (_s != HSA_STATUS_SUCCESS) { const char* m = nullptr; hsa_status_string(_s, &m); \
    fprintf(stderr, "%s failed: %s (line %d)\n", #e, m ? m : "?", __LINE__); exit(2); } } while (0)

static hsa_agent_t g_gpu{}, g_cpu{};
static hsa_amd_memory_pool_t g_gpu_pool{}, g_gpu_fine{}, g_host_kernarg{};
static bool g_have_gpu = false, g_have_cpu = false;
static hsa_status_t agent_cb(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t gpu_pool_cb(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t fl; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
    bool alloc; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_gpu_pool.handle) g_gpu_pool = p;
    if (alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED) && !g_gpu_fine.handle) g_gpu_fine = p;
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t cpu_pool_cb(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t fl; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
    if ((fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_host_kernarg.handle) g_host_kernarg = p;
    return HSA_STATUS_SUCCESS;
}
struct Kernel { uint64_t object; uint32_t kernarg, group, priv; };
static Kernel get_kernel(hsa_executable_t exe, const char* name) {
    hsa_executable_symbol_t sym;
    HCK(hsa_executable_get_symbol_by_name(exe, (std::string(name) + ".kd").c_str(), &g_gpu, &sym));
    Kernel k{};
    HCK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    HCK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
    HCK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    HCK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    return k;
}
struct Q {
    hsa_queue_t* q = nullptr; uint64_t widx = 0; hsa_signal_t done{};
    void dispatch(const Kernel& k, uint32_t grid_wg, void* kernarg, int scope, hsa_signal_t sig) {
        auto* p = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(q->base_address) + (widx & (q->size - 1));
        p->workgroup_size_x = 64; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
        p->grid_size_x = grid_wg * 64; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = k.priv; p->group_segment_size = k.group;
        p->kernel_object = k.object; p->kernarg_address = kernarg; p->reserved2 = 0; p->completion_signal = sig;
        const uint16_t hdr = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                        (scope << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (scope << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)hdr | ((uint32_t)(1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS) << 16), __ATOMIC_RELEASE);
        ++widx;
    }
    void barrier_value(hsa_signal_t sig, hsa_signal_value_t below) {     // wait until sig < below
        auto* p = reinterpret_cast<hsa_amd_barrier_value_packet_t*>(q->base_address) + (widx & (q->size - 1));
        std::memset(reinterpret_cast<char*>(p) + 4, 0, sizeof(*p) - 4);
        p->signal = sig; p->value = below; p->mask = ~(hsa_signal_value_t)0; p->cond = HSA_SIGNAL_CONDITION_LT;
        const uint16_t hdr = (uint16_t)((HSA_PACKET_TYPE_VENDOR_SPECIFIC << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER));
        const uint32_t first = (uint32_t)hdr | ((uint32_t)HSA_AMD_PACKET_TYPE_BARRIER_VALUE << 16);
        __atomic_store_n(reinterpret_cast<uint32_t*>(p), first, __ATOMIC_RELEASE);
        ++widx;
    }
    void finish() {                                                       // barrier-AND with the done signal, doorbell
        auto* b = reinterpret_cast<hsa_barrier_and_packet_t*>(q->base_address) + (widx & (q->size - 1));
        std::memset(reinterpret_cast<char*>(b) + 4, 0, sizeof(*b) - 4);
        hsa_signal_store_relaxed(done, 1);
        b->completion_signal = done;
        __atomic_store_n(reinterpret_cast<uint32_t*>(b), (uint32_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER)), __ATOMIC_RELEASE);
        ++widx;
    }
    void ring() { hsa_queue_store_write_index_screlease(q, widx); hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(widx - 1)); }
    bool wait() { return hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 3000000000ull, HSA_WAIT_STATE_ACTIVE) < 1; }
};
struct SyncArgs { unsigned long long* mine; unsigned long long* peer; unsigned long long seq, timeout; };
struct WorkArgs { int* p; };

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "build_variants/handover_kernels.hsaco";
    HCK(hsa_init());
    HCK(hsa_iterate_agents(agent_cb, nullptr));
    if (!g_have_gpu || !g_have_cpu) { fprintf(stderr, "no GPU agent\n"); return 2; }
    HCK(hsa_amd_agent_iterate_memory_pools(g_gpu, gpu_pool_cb, nullptr));
    HCK(hsa_amd_agent_iterate_memory_pools(g_cpu, cpu_pool_cb, nullptr));
    std::ifstream f(path, std::ios::binary);
    std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (blob.empty()) { fprintf(stderr, "cannot read %s\n", path); return 2; }
    hsa_code_object_reader_t rd; hsa_executable_t exe;
    HCK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &rd));
    HCK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HCK(hsa_executable_load_agent_code_object(exe, g_gpu, rd, nullptr, nullptr));
    HCK(hsa_executable_freeze(exe, nullptr));
    const Kernel k_work = get_kernel(exe, "k_work"), k_sync = get_kernel(exe, "k_sync");

    const int N = 1000;                       // hand-overs per run; packets per queue <= 2 N + 1
    Q q[2];
    for (auto& x : q) { HCK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &x.q)); HCK(hsa_signal_create(1, 0, nullptr, &x.done)); x.widx = hsa_queue_load_write_index_relaxed(x.q); }
    char* ka = nullptr;                       // kernel arguments in host kernarg memory (this benchmark's kernels are tiny: the cost is the same for all variants)
    HCK(hsa_amd_memory_pool_allocate(g_host_kernarg, (size_t)4 * N * 64, 0, (void**)&ka));
    HCK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, ka));
    unsigned long long* flags = nullptr;      // [2] in fine-grained device memory, 256 bytes apart
    HCK(hsa_amd_memory_pool_allocate(g_gpu_fine.handle ? g_gpu_fine : g_gpu_pool, 4096, 0, (void**)&flags));
    HCK(hsa_amd_agents_allow_access(1, &g_cpu, nullptr, flags));
    hsa_signal_t cs[2];
    const hsa_signal_value_t BIG = (hsa_signal_value_t)1 << 40;
    for (auto& s : cs) HCK(hsa_signal_create(BIG, 0, nullptr, &s));
    hsa_signal_t none{0};

    for (int scope : {HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_SYSTEM}) {
        double res[3] = {0, 0, 0};
        for (int variant = 0; variant < 3; ++variant) {
            double best = 1e30;
            for (int rep = 0; rep < 4; ++rep) {
                HCK(hsa_amd_memory_fill(flags, 0u, 1024));
                for (auto& s : cs) hsa_signal_store_relaxed(s, BIG);
                size_t slot = 0;
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; ++i)
                    for (int r = 0; r < 2; ++r) {
                        WorkArgs* wa = reinterpret_cast<WorkArgs*>(ka + (slot++) * 64); wa->p = nullptr;
                        q[r].dispatch(k_work, 64, wa, scope, variant == 2 ? cs[r] : none);
                        if (variant == 1) {
                            SyncArgs* sa = reinterpret_cast<SyncArgs*>(ka + (slot++) * 64);
                            sa->mine = flags + 32 * r; sa->peer = flags + 32 * (r ^ 1); sa->seq = (unsigned long long)i + 1; sa->timeout = 200000000ull;
                            q[r].dispatch(k_sync, 1, sa, HSA_FENCE_SCOPE_NONE, none);
                        } else if (variant == 2) {
                            q[r].barrier_value(cs[r ^ 1], BIG - i);          // the peer's i-th work kernel has finished: its signal is BIG - (i + 1)
                        }
                    }
                for (auto& x : q) { x.finish(); x.ring(); }
                bool ok = true;
                for (auto& x : q) ok = x.wait() && ok;
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (!ok) { printf("variant %d: TIMEOUT (queue stuck)\n", variant); return 3; }
                best = us < best ? us : best;
            }
            res[variant] = best / N;
        }
        printf("%s-scope fences on the work kernels, two queues on one GPU, per iteration: A no hand-over %.2f us | B announce + poll kernel %.2f us (hand-over %.2f) | "
               "C completion signal + barrier-value packet %.2f us (hand-over %.2f)\n", scope == HSA_FENCE_SCOPE_AGENT ? "agent" : "system", res[0], res[1], res[1] - res[0], res[2],
               res[2] - res[0]);
    }
    return 0;
}
