// Microbenchmark (diagnostic): dependent kernel dispatches written as AQL packets into a user-mode HSA queue by this program
// itself -- no HIP launch call per kernel.  Questions: (1) what does a dependent dispatch cost when the host is not in the loop
// (all packets written, one doorbell)? (2) how much of the kernel boundary is the acquire / release fence of the packet header
// (system, agent, none)? (3) is a fence-less boundary correct across XCDs when the kernel uses coherent (sc1) accesses?
//   hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 -o build_variants/aql_direct_kernels.hsaco tools/micro/aql_direct_kernels.hip
//   g++ -O2 -std=c++17 -I/opt/rocm/include -o build_variants/aql_direct tools/micro/aql_direct.cpp -L/opt/rocm/lib -lhsa-runtime64 -lamdhip64 -Wl,-rpath,/opt/rocm/lib
//   ./build_variants/aql_direct build_variants/aql_direct_kernels.hsaco
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include <x86intrin.h>

#define HCK(e) do { hsa_status_t _s = (e); if (_s != HSA_STATUS_SUCCESS) { const char* m = nullptr; hsa_status_string(_s, &m); \
    fprintf(stderr, "%s failed: %s (line %d)\n", #e, m ? m : "?", __LINE__); exit(2); } } while (0)

static hsa_agent_t g_gpu{}, g_cpu{};
static hsa_amd_memory_pool_t g_gpu_pool{}, g_gpu_fine_pool{}, g_gpu_ext_fine_pool{}, g_host_kernarg_pool{}, g_host_fine_pool{};
static bool g_have_gpu = false, g_have_cpu = false;

static hsa_status_t agent_cb(hsa_agent_t a, void*) {
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
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
    if (alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED) && !g_gpu_fine_pool.handle) g_gpu_fine_pool = p;
    if (alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_EXTENDED_SCOPE_FINE_GRAINED) && !g_gpu_ext_fine_pool.handle) g_gpu_ext_fine_pool = p;
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t cpu_pool_cb(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t fl; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
    if ((fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_host_kernarg_pool.handle) g_host_kernarg_pool = p;
    if ((fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED) && !g_host_fine_pool.handle) g_host_fine_pool = p;
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

static hsa_queue_t* g_q = nullptr;
static uint64_t g_widx = 0;
static hsa_signal_t g_done{};

static inline uint16_t header(int barrier, int acq, int rel) {
    return (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (barrier << HSA_PACKET_HEADER_BARRIER) |
                      (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
}
static inline void write_packet(const Kernel& k, uint32_t grid_wg, void* kernarg, uint16_t hdr, hsa_signal_t sig, uint32_t block = 64) {
    auto* base = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(g_q->base_address);
    hsa_kernel_dispatch_packet_t* p = base + (g_widx & (g_q->size - 1));
    p->workgroup_size_x = (uint16_t)block; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
    p->grid_size_x = grid_wg * block; p->grid_size_y = 1; p->grid_size_z = 1;
    p->private_segment_size = k.priv; p->group_segment_size = k.group;
    p->kernel_object = k.object; p->kernarg_address = kernarg; p->reserved2 = 0; p->completion_signal = sig;
    const uint32_t first = (uint32_t)hdr | ((uint32_t)(3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS) << 16);
    __atomic_store_n(reinterpret_cast<uint32_t*>(p), first, __ATOMIC_RELEASE);
    ++g_widx;
}
static inline void ring() {
    hsa_queue_store_write_index_screlease(g_q, g_widx);
    hsa_signal_store_screlease(g_q->doorbell_signal, (hsa_signal_value_t)(g_widx - 1));
}
static bool wait_done() {
    const hsa_signal_value_t v = hsa_signal_wait_scacquire(g_done, HSA_SIGNAL_CONDITION_LT, 1, 1000000000ull /* ticks; bounded */, HSA_WAIT_STATE_ACTIVE);
    return v < 1;
}

struct ChainArgs { double* x; uint32_t nb, shift; };
struct GatherArgs { void* tab; uint32_t mask, rows, shift; };
struct EmptyArgs { int* p; };

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "build_variants/aql_direct_kernels.hsaco";
    HCK(hsa_init());
    HCK(hsa_iterate_agents(agent_cb, nullptr));
    if (!g_have_gpu || !g_have_cpu) { fprintf(stderr, "no GPU agent\n"); return 2; }
    HCK(hsa_amd_agent_iterate_memory_pools(g_gpu, gpu_pool_cb, nullptr));
    HCK(hsa_amd_agent_iterate_memory_pools(g_cpu, cpu_pool_cb, nullptr));
    char name[64] = {0}; hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, name);
    hsa_amd_memory_pool_access_t acc;
    HCK(hsa_amd_agent_memory_pool_get_info(g_cpu, g_gpu_pool, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &acc));
    printf("agent %s; CPU access to the GPU pool: %s\n", name, acc == HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED ? "never" : "possible (large BAR)");
    const bool bar = acc != HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;

    std::ifstream f(path, std::ios::binary);
    std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (blob.empty()) { fprintf(stderr, "cannot read %s\n", path); return 2; }
    hsa_code_object_reader_t rd; hsa_executable_t exe;
    HCK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &rd));
    HCK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HCK(hsa_executable_load_agent_code_object(exe, g_gpu, rd, nullptr, nullptr));
    HCK(hsa_executable_freeze(exe, nullptr));
    const Kernel k_empty = get_kernel(exe, "k_empty"), k_chain = get_kernel(exe, "k_chain"), k_chain_coh = get_kernel(exe, "k_chain_coherent"),
                 k_gather = get_kernel(exe, "k_gather");

    HCK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &g_q));
    HCK(hsa_signal_create(1, 0, nullptr, &g_done));

    const int N = 2000;
    // kernarg slots: one per packet of a run; in device memory (host writes through the BAR) or in host kernarg memory
    char *ka_dev = nullptr, *ka_host = nullptr;
    HCK(hsa_amd_memory_pool_allocate(g_host_kernarg_pool, (size_t)N * 64, 0, (void**)&ka_host));
    HCK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, ka_host));
    if (bar) {
        HCK(hsa_amd_memory_pool_allocate(g_gpu_pool, (size_t)N * 64, 0, (void**)&ka_dev));
        HCK(hsa_amd_agents_allow_access(1, &g_cpu, nullptr, ka_dev));
    }
    const uint32_t NB = 4096;                                   // 4096 blocks of 64 doubles = 2 MB
    double* x; HCK(hsa_amd_memory_pool_allocate(g_gpu_pool, (size_t)NB * 64 * 8, 0, (void**)&x));
    const uint32_t NT = 1u << 19;                               // 8 MB table of 16-byte pieces (cfg2's state matrix is 6.5 MB)
    void* tab; HCK(hsa_amd_memory_pool_allocate(g_gpu_pool, (size_t)NT * 16, 0, (void**)&tab));
    HCK(hsa_amd_memory_fill(tab, 1u, (size_t)NT * 4));
    std::vector<double> hx((size_t)NB * 64);

    struct Variant { const char* name; int barrier, acq, rel; };
    const Variant variants[] = {{"barrier, acquire/release SYSTEM", 1, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM},
                                {"barrier, acquire/release AGENT ", 1, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT},
                                {"barrier, acquire AGENT, rel NONE", 1, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_NONE},
                                {"barrier, acquire NONE, rel AGENT", 1, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_AGENT},
                                {"barrier, no fences             ", 1, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE},
                                {"no barrier, no fences (overlap)", 0, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE}};
    printf("%-34s %-10s %-22s %-12s %10s  %s\n", "packet header", "kernarg", "kernel", "doorbell", "us/launch", "check");
    const bool part1 = !(argc > 2 && (!strcmp(argv[2], "mem") || !strcmp(argv[2], "p4") || !strcmp(argv[2], "p5")));
    const bool full = argc > 2 && !strcmp(argv[2], "full");
    for (int where = (bar && !full) ? 1 : 0; part1 && where < (bar ? 2 : 1); ++where) {
        char* ka = where ? ka_dev : ka_host;
        for (const Variant& v : variants) {
            for (int kind = 0; kind < 6; ++kind) {
                // kind: 0 empty 4096 wg, 1 empty 512 wg, 2 chain (plain), 3 chain (coherent accesses), 4 gather 4096 wg x 7 rows, 5 gather 512 wg x 7 rows
                for (int per_packet_doorbell = 0; per_packet_doorbell < 2; ++per_packet_doorbell) {
                    if (per_packet_doorbell && !(kind == 0 || kind == 4)) continue;
                    if ((kind == 2 || kind == 3)) { HCK(hsa_amd_memory_fill(x, 0u, (size_t)NB * 64 * 2)); }
                    double best = 1e30;
                    int reps = 3;
                    uint32_t total_launches = 0;
                    for (int rep = 0; rep < reps; ++rep) {
                        hsa_signal_store_relaxed(g_done, 1);
                        for (int i = 0; i < N; ++i) {        // arguments first
                            char* slot = ka + (size_t)i * 64;
                            if (kind <= 1) { EmptyArgs a{nullptr}; memcpy(slot, &a, sizeof(a)); }
                            else if (kind <= 3) { ChainArgs a{x, NB, total_launches + (uint32_t)i * 7u}; memcpy(slot, &a, sizeof(a)); }
                            else { GatherArgs a{tab, NT - 1, 7u, total_launches + (uint32_t)i}; memcpy(slot, &a, sizeof(a)); }
                        }
                        _mm_sfence();
                        if (where) { volatile char sink = ka[(size_t)(N - 1) * 64]; (void)sink; }   // read back: the posted writes have landed
                        const Kernel& k = kind <= 1 ? k_empty : (kind == 2 ? k_chain : (kind == 3 ? k_chain_coh : k_gather));
                        const uint32_t wg = (kind == 1 || kind == 5) ? 512u : 4096u;
                        const uint16_t hdr = header(v.barrier, v.acq, v.rel);
                        const uint16_t hdr_last = header(1, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM);
                        const auto t0 = std::chrono::high_resolution_clock::now();
                        for (int i = 0; i < N; ++i) {
                            const bool last = i == N - 1;
                            write_packet(k, wg, ka + (size_t)i * 64, last ? hdr_last : hdr, last ? g_done : hsa_signal_t{0});
                            if (per_packet_doorbell) ring();
                        }
                        if (!per_packet_doorbell) ring();
                        if (!wait_done()) { fprintf(stderr, "timeout waiting for the queue (variant %s kind %d)\n", v.name, kind); return 3; }
                        const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / N;
                        best = us < best ? us : best;
                        total_launches += (uint32_t)N;
                    }
                    const char* check = "";
                    if (kind == 2 || kind == 3) {
                        HCK(hsa_memory_copy(hx.data(), x, hx.size() * 8));
                        size_t bad = 0;
                        for (double e : hx) bad += e != (double)total_launches;
                        static char buf[64];
                        snprintf(buf, sizeof buf, bad ? "WRONG: %zu of %zu elements" : "x == launches everywhere", bad, hx.size());
                        check = buf;
                    }
                    static const char* kn[] = {"empty 4096x64", "empty 512x64", "chain 4096x64 plain", "chain 4096x64 coherent", "gather 4096x64 7 rows", "gather 512x64 7 rows"};
                    printf("%-34s %-10s %-22s %-12s %10.2f  %s\n", v.name, where ? "device" : "host", kn[kind], per_packet_doorbell ? "per packet" : "once", best, check);
                    fflush(stdout);
                }
            }
        }
    }

    const bool only4 = argc > 2 && (!strcmp(argv[2], "p4") || !strcmp(argv[2], "p5"));
    // ---- part 2: the data in other kinds of device memory (plain kernels): does memory the L2 does not keep non-coherent
    // copies of make a fence-less boundary correct, and what do its reads cost?
    if (bar && !only4) {
        struct Mem { const char* name; hsa_amd_memory_pool_t pool; uint32_t flags; };
        const Mem mems[] = {{"coarse-grained", g_gpu_pool, 0}, {"coarse + UNCACHED flag", g_gpu_pool, HSA_AMD_MEMORY_POOL_UNCACHED_FLAG},
                            {"fine-grained", g_gpu_fine_pool, 0}, {"fine + UNCACHED flag", g_gpu_fine_pool, HSA_AMD_MEMORY_POOL_UNCACHED_FLAG},
                            {"ext-scope fine-grained", g_gpu_ext_fine_pool, 0}};
        printf("\n%-24s %-34s %-22s %10s  %s\n", "data memory", "packet header", "kernel", "us/launch", "check");
        for (const Mem& m : mems) {
            if (!m.pool.handle) { printf("%-24s (no such pool)\n", m.name); continue; }
            double* mx = nullptr; void* mtab = nullptr;
            if (hsa_amd_memory_pool_allocate(m.pool, (size_t)NB * 64 * 8, m.flags, (void**)&mx) != HSA_STATUS_SUCCESS ||
                hsa_amd_memory_pool_allocate(m.pool, (size_t)NT * 16, m.flags, (void**)&mtab) != HSA_STATUS_SUCCESS) { printf("%-24s (allocation refused)\n", m.name); continue; }
            HCK(hsa_amd_memory_fill(mtab, 1u, (size_t)NT * 4));
            for (int vi : {1, 4}) {
                const Variant& v = variants[vi];
                for (int kind : {2, 3, 4, 5}) {
                    if (kind == 2 || kind == 3) HCK(hsa_amd_memory_fill(mx, 0u, (size_t)NB * 64 * 2));
                    double best = 1e30; uint32_t total_launches = 0;
                    for (int rep = 0; rep < 3; ++rep) {
                        hsa_signal_store_relaxed(g_done, 1);
                        for (int i = 0; i < N; ++i) {
                            char* slot = ka_dev + (size_t)i * 64;
                            if (kind <= 3) { ChainArgs a{mx, NB, total_launches + (uint32_t)i * 7u}; memcpy(slot, &a, sizeof(a)); }
                            else { GatherArgs a{mtab, NT - 1, 7u, total_launches + (uint32_t)i}; memcpy(slot, &a, sizeof(a)); }
                        }
                        _mm_sfence();
                        { volatile char sink = ka_dev[(size_t)(N - 1) * 64]; (void)sink; }
                        const Kernel& k = kind == 2 ? k_chain : (kind == 3 ? k_chain_coh : k_gather);
                        const uint32_t wg = kind == 5 ? 512u : 4096u;
                        const auto t0 = std::chrono::high_resolution_clock::now();
                        for (int i = 0; i < N; ++i) {
                            const bool last = i == N - 1;
                            write_packet(k, wg, ka_dev + (size_t)i * 64, last ? header(1, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM) : header(v.barrier, v.acq, v.rel),
                                         last ? g_done : hsa_signal_t{0});
                        }
                        ring();
                        if (!wait_done()) { fprintf(stderr, "timeout (memory %s)\n", m.name); return 3; }
                        const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / N;
                        best = us < best ? us : best;
                        total_launches += (uint32_t)N;
                    }
                    char buf[64] = "";
                    if (kind <= 3) {
                        HCK(hsa_memory_copy(hx.data(), mx, hx.size() * 8));
                        size_t bad = 0;
                        for (double e : hx) bad += e != (double)total_launches;
                        snprintf(buf, sizeof buf, bad ? "WRONG: %zu of %zu elements" : "x == launches everywhere", bad, hx.size());
                    }
                    static const char* kn[] = {"", "", "chain 4096x64 plain", "chain 4096x64 coherent", "gather 4096x64 7 rows", "gather 512x64 7 rows"};
                    printf("%-24s %-34s %-22s %10.2f  %s\n", m.name, v.name, kn[kind], best, buf);
                    fflush(stdout);
                }
            }
            hsa_amd_memory_pool_free(mx); hsa_amd_memory_pool_free(mtab);
        }
    }

    // ---- part 3: the data allocated by HIP (the product's allocator), kernels still through this program's own queue
    if (bar && !only4) {
        struct HMem { const char* name; int kind; };
        const HMem hm[] = {{"hipMalloc", 0}, {"hipExtMalloc Finegrained", 1}, {"hipExtMalloc Uncached", 2}};
        printf("\n%-26s %-34s %-22s %10s  %s\n", "data memory (HIP)", "packet header", "kernel", "us/launch", "check");
        for (const HMem& m : hm) {
            double* mx = nullptr; void* mtab = nullptr;
            hipError_t e1, e2;
            if (m.kind == 0) { e1 = hipMalloc((void**)&mx, (size_t)NB * 64 * 8); e2 = hipMalloc(&mtab, (size_t)NT * 16); }
            else { const unsigned fl = m.kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached;
                   e1 = hipExtMallocWithFlags((void**)&mx, (size_t)NB * 64 * 8, fl); e2 = hipExtMallocWithFlags(&mtab, (size_t)NT * 16, fl); }
            if (e1 != hipSuccess || e2 != hipSuccess) { printf("%-26s (allocation refused)\n", m.name); continue; }
            hipMemset(mtab, 1, (size_t)NT * 16);
            for (int vi : {1, 4}) {
                const Variant& v = variants[vi];
                for (int kind : {2, 4, 5}) {
                    if (kind == 2) { hipMemset(mx, 0, (size_t)NB * 64 * 8); }
                    hipDeviceSynchronize();
                    double best = 1e30; uint32_t total_launches = 0;
                    for (int rep = 0; rep < 3; ++rep) {
                        hsa_signal_store_relaxed(g_done, 1);
                        for (int i = 0; i < N; ++i) {
                            char* slot = ka_dev + (size_t)i * 64;
                            if (kind <= 3) { ChainArgs a{mx, NB, total_launches + (uint32_t)i * 7u}; memcpy(slot, &a, sizeof(a)); }
                            else { GatherArgs a{mtab, NT - 1, 7u, total_launches + (uint32_t)i}; memcpy(slot, &a, sizeof(a)); }
                        }
                        _mm_sfence();
                        { volatile char sink = ka_dev[(size_t)(N - 1) * 64]; (void)sink; }
                        const Kernel& k = kind == 2 ? k_chain : k_gather;
                        const uint32_t wg = kind == 5 ? 512u : 4096u;
                        const auto t0 = std::chrono::high_resolution_clock::now();
                        for (int i = 0; i < N; ++i) {
                            const bool last = i == N - 1, first = i == 0;
                            write_packet(k, wg, ka_dev + (size_t)i * 64, (last || first) ? header(1, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM) : header(v.barrier, v.acq, v.rel),
                                         last ? g_done : hsa_signal_t{0});
                        }
                        ring();
                        if (!wait_done()) { fprintf(stderr, "timeout (memory %s)\n", m.name); return 3; }
                        const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / N;
                        best = us < best ? us : best;
                        total_launches += (uint32_t)N;
                    }
                    char buf[64] = "";
                    if (kind == 2) {
                        hipMemcpy(hx.data(), mx, hx.size() * 8, hipMemcpyDeviceToHost);
                        size_t bad = 0;
                        for (double e : hx) bad += e != (double)total_launches;
                        snprintf(buf, sizeof buf, bad ? "WRONG: %zu of %zu elements" : "x == launches everywhere", bad, hx.size());
                    }
                    static const char* kn[] = {"", "", "chain 4096x64 plain", "", "gather 4096x64 7 rows", "gather 512x64 7 rows"};
                    printf("%-26s %-34s %-22s %10.2f  %s\n", m.name, v.name, kn[kind], best, buf);
                    fflush(stdout);
                }
            }
            hipFree(mx); hipFree(mtab);
        }
    }

    // ---- part 4: what does a THIRD, tiny kernel cost behind two heavy ones (the burn-in generation: update, update, cr_adapt)?
    if (bar && !(argc > 2 && !strcmp(argv[2], "p5"))) {
        const Kernel k_e1024 = get_kernel(exe, "k_empty_1024");
        printf("\n%-60s %10s\n", "chain of dependent dispatches (acquire-only packets), per round", "us/round");
        for (int pattern = 0; pattern < 5; ++pattern) {
            // 0: gather, gather   1: gather, gather, empty 1x64   2: gather, gather, empty 1x1024 (10 KB LDS)   3: empty 1x1024 alone  4: gather, gather, empty 16x64
            const int per_round = pattern == 0 ? 2 : (pattern == 3 ? 1 : 3);
            const int rounds = 600;
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                hsa_signal_store_relaxed(g_done, 1);
                for (int i = 0; i < rounds * per_round; ++i) {
                    char* slot = ka_dev + (size_t)i * 64;
                    GatherArgs a{tab, NT - 1, 7u, (uint32_t)i}; memcpy(slot, &a, sizeof(a));
                }
                _mm_sfence();
                { volatile char sink = ka_dev[(size_t)(rounds * per_round - 1) * 64]; (void)sink; }
                const uint16_t hdr = header(1, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_NONE);
                const auto t0 = std::chrono::high_resolution_clock::now();
                int i = 0;
                for (int r = 0; r < rounds; ++r) {
                    for (int j = 0; j < per_round; ++j, ++i) {
                        const bool last = i == rounds * per_round - 1;
                        const bool tiny = (pattern != 0 && pattern != 3 && j == 2) || pattern == 3;
                        const uint16_t h = last ? header(1, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM) : hdr;
                        if (!tiny) write_packet(k_gather, 4096, ka_dev + (size_t)i * 64, h, last ? g_done : hsa_signal_t{0});
                        else if (pattern == 1) write_packet(k_empty, 1, ka_dev + (size_t)i * 64, h, last ? g_done : hsa_signal_t{0});
                        else if (pattern == 4) write_packet(k_empty, 16, ka_dev + (size_t)i * 64, h, last ? g_done : hsa_signal_t{0});
                        else write_packet(k_e1024, 1, ka_dev + (size_t)i * 64, h, last ? g_done : hsa_signal_t{0}, 1024);
                    }
                }
                ring();
                if (!wait_done()) { fprintf(stderr, "timeout (part 4)\n"); return 3; }
                const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / rounds;
                best = us < best ? us : best;
            }
            static const char* pn[] = {"gather 4096x64, gather 4096x64", "gather, gather, empty 1x64", "gather, gather, empty 1x1024 with 10 KB LDS", "empty 1x1024 with 10 KB LDS alone",
                                       "gather, gather, empty 16x64"};
            printf("%-60s %10.2f\n", pn[pattern], best);
            fflush(stdout);
        }
        // the sampler's own reduction kernel (argv[3] = the library's device code object), N = 0: what does THIS kernel cost as the third dispatch?
        if (argc > 3) {
            std::ifstream f2(argv[3], std::ios::binary);
            std::vector<char> blob2((std::istreambuf_iterator<char>(f2)), std::istreambuf_iterator<char>());
            hsa_code_object_reader_t rd2; hsa_executable_t exe2;
            HCK(hsa_code_object_reader_create_from_memory(blob2.data(), blob2.size(), &rd2));
            HCK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe2));
            HCK(hsa_executable_load_agent_code_object(exe2, g_gpu, rd2, nullptr, nullptr));
            HCK(hsa_executable_freeze(exe2, nullptr));
            const Kernel k_adapt = get_kernel(exe2, "_ZN3bpm15cr_adapt_kernelENS_6LayoutEjjPdS1_Pjj");
            printf("cr_adapt_kernel: kernarg %u B, LDS %u B, scratch %u B\n", k_adapt.kernarg, k_adapt.group, k_adapt.priv);
            struct AdaptArgs { void* G; uint64_t blk; uint32_t n_local, ld, dim, world, magic, pad; uint32_t N, n_cr; void* cr_state; void* part; void* ticket; uint32_t span; };
            static_assert(sizeof(AdaptArgs) == 88 || sizeof(AdaptArgs) == 80, "layout");
            for (int pattern = 0; pattern < 3; ++pattern) {     // 0: adapt alone, 1: gather, gather, adapt(N=0), 2: gather, gather, adapt(N=8192 over the table)
                const int per_round = pattern == 0 ? 1 : 3, rounds = 300;
                double best = 1e30;
                for (int rep = 0; rep < 3; ++rep) {
                    hsa_signal_store_relaxed(g_done, 1);
                    for (int i = 0; i < rounds * per_round; ++i) {
                        char* slot = ka_dev + (size_t)i * 128;
                        const bool tiny = pattern == 0 || (i % 3) == 2;
                        if (tiny) {
                            AdaptArgs a{}; a.G = tab; a.blk = 8192ull * 102; a.n_local = 8192; a.ld = 100; a.dim = 100; a.world = 1; a.magic = (uint32_t)((1ull << 32) / 8192) + 1u;
                            a.N = pattern == 2 ? 8192u : 0u; a.n_cr = 3; a.cr_state = x; a.part = nullptr; a.ticket = nullptr; a.span = 0;
                            memcpy(slot, &a, sizeof(a));
                            struct { uint32_t bc[3]; uint16_t gs[3], rem[3]; } hid{{1, 1, 1}, {1024, 1, 1}, {0, 0, 0}};
                            memcpy(slot + ((sizeof(a) + 7) & ~size_t(7)), &hid, sizeof(hid));
                        } else { GatherArgs a{tab, NT - 1, 7u, (uint32_t)i}; memcpy(slot, &a, sizeof(a)); }
                    }
                    _mm_sfence();
                    { volatile char sink = ka_dev[(size_t)(rounds * per_round - 1) * 128]; (void)sink; }
                    const uint16_t hdr = header(1, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_NONE);
                    const auto t0 = std::chrono::high_resolution_clock::now();
                    for (int i = 0; i < rounds * per_round; ++i) {
                        const bool last = i == rounds * per_round - 1;
                        const bool tiny = pattern == 0 || (i % 3) == 2;
                        const uint16_t h = last ? header(1, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM) : hdr;
                        if (tiny) write_packet(k_adapt, 1, ka_dev + (size_t)i * 128, h, last ? g_done : hsa_signal_t{0}, 1024);
                        else write_packet(k_gather, 4096, ka_dev + (size_t)i * 128, h, last ? g_done : hsa_signal_t{0});
                    }
                    ring();
                    if (!wait_done()) { fprintf(stderr, "timeout (part 4b)\n"); return 3; }
                    const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / rounds;
                    best = us < best ? us : best;
                }
                static const char* pn2[] = {"cr_adapt_kernel (N = 0) alone", "gather, gather, cr_adapt_kernel (N = 0)", "gather, gather, cr_adapt_kernel (N = 8192)"};
                printf("%-60s %10.2f\n", pn2[pattern], best);
                fflush(stdout);
            }
        }
    }

    // ---- part 5: does what launch i brings into an XCD's L2 survive the kernel boundary (prefetch of the next launch's records)?
    if (bar && argc > 2 && !strcmp(argv[2], "p5")) {
        const Kernel k_slice = get_kernel(exe, "k_slice");
        struct SliceArgs { void* rec; void* data; void* out; uint32_t slice, n_per_slice, mask, prefetch; };
        const uint32_t NPS = 4096, SLICES = 600;
        struct Mem { const char* name; hsa_amd_memory_pool_t pool; };
        const Mem mems[] = {{"records in ordinary memory", g_gpu_pool}, {"records in ext-scope fine-grained memory", g_gpu_ext_fine_pool}};
        printf("\n%-44s %-28s %-9s %10s\n", "record table", "packet header", "prefetch", "us/launch");
        for (const Mem& m : mems) {
            if (!m.pool.handle) continue;
            void *rec = nullptr, *out = nullptr;
            HCK(hsa_amd_memory_pool_allocate(m.pool, (size_t)(SLICES + 2) * NPS * 64, 0, &rec));
            HCK(hsa_amd_memory_pool_allocate(g_gpu_pool, (size_t)NPS * 16, 0, &out));
            HCK(hsa_amd_memory_fill(rec, 12345u, (size_t)(SLICES + 2) * NPS * 16));
            for (int vi : {2, 4}) {      // acquire only / no fences
                const Variant& v = variants[vi];
                for (uint32_t pf = 0; pf < 2; ++pf) {
                    double best = 1e30;
                    for (int rep = 0; rep < 3; ++rep) {
                        hsa_signal_store_relaxed(g_done, 1);
                        for (uint32_t i = 0; i < SLICES; ++i) {
                            SliceArgs a{rec, tab, out, i, NPS, NT - 1, pf};
                            memcpy(ka_dev + (size_t)i * 64, &a, sizeof(a));
                        }
                        _mm_sfence();
                        { volatile char sink = ka_dev[(size_t)(SLICES - 1) * 64]; (void)sink; }
                        const auto t0 = std::chrono::high_resolution_clock::now();
                        for (uint32_t i = 0; i < SLICES; ++i) {
                            const bool last = i == SLICES - 1;
                            write_packet(k_slice, NPS, ka_dev + (size_t)i * 64, last ? header(1, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM) : header(v.barrier, v.acq, v.rel),
                                         last ? g_done : hsa_signal_t{0});
                        }
                        ring();
                        if (!wait_done()) { fprintf(stderr, "timeout (part 5)\n"); return 3; }
                        const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / SLICES;
                        best = us < best ? us : best;
                    }
                    printf("%-44s %-28s %-9u %10.2f\n", m.name, v.name, pf, best);
                    fflush(stdout);
                }
            }
            hsa_amd_memory_pool_free(rec); hsa_amd_memory_pool_free(out);
        }
    }
    hsa_queue_destroy(g_q);
    hsa_shut_down();
    return 0;
}
