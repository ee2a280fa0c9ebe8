// What is the 1.8 us of a dependent launch made of?  Raw AQL dispatches on a queue of our own (HSA runtime, no HIP): a chain of N dependent
// kernels with the packet header's barrier bit and acquire / release fence scopes varied.  HIP (graph or stream) always emits
// barrier + agent- or system-scope fences between the kernels of one stream; this probe asks what the barrier alone costs, and whether
// a chain whose data crosses the boundary through write-through stores / L2-bypassing loads stays correct without the fences.
//
//   hipcc --offload-arch=gfx950 --offload-device-only --no-gpu-bundle-output -O3 perf/aql_fence_probe_kernels.hip -o perf/aql_fence_probe.hsaco
//   g++ -O2 -I/opt/rocm/include perf/aql_fence_probe.cpp -L/opt/rocm/lib -lhsa-runtime64 -Wl,-rpath,/opt/rocm/lib -o perf/aql_fence_probe.bin
//   perf/aql_fence_probe.bin perf/aql_fence_probe.hsaco [chain length] [device-kernarg|host-kernarg] [doorbell-once|doorbell-per-packet]
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char *m = nullptr; hsa_status_string(s_, &m); \
    fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m ? m : "?"); exit(2); } } while (0)

static hsa_agent_t g_gpu{}, g_cpu{};
static hsa_amd_memory_pool_t g_dev_pool{}, g_kernarg_pool{}, g_fine_pool{};
static bool g_have_gpu = false, g_have_cpu = false;

static hsa_status_t on_agent(hsa_agent_t a, void *) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_gpu_pool(hsa_amd_memory_pool_t p, void *) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t fl; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
    bool alloc; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_dev_pool.handle) g_dev_pool = p;
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_cpu_pool(hsa_amd_memory_pool_t p, void *) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t fl; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
    if ((fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_kernarg_pool.handle) g_kernarg_pool = p;
    if ((fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED) && !g_fine_pool.handle) g_fine_pool = p;
    return HSA_STATUS_SUCCESS;
}

struct Kernel { uint64_t object; uint32_t kernarg, group, priv; };
static Kernel get_kernel(hsa_executable_t ex, const char *name) {
    hsa_executable_symbol_t sym; CK(hsa_executable_get_symbol_by_name(ex, name, &g_gpu, &sym));
    Kernel k{};
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    return k;
}

struct ChainArgs { uint32_t *a; uint32_t *b; uint32_t nb; uint32_t pad; };   // k_chain(const uint32_t* src, uint32_t* dst, uint32_t nb)

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s code_object.hsaco [chain_length]\n", argv[0]); return 2; }
    const int N = argc > 2 ? atoi(argv[2]) : 2000, GRID = 256, BLOCK = 1024;
    const bool per_packet_doorbell = argc > 4 && !strcmp(argv[4], "doorbell-per-packet");
    CK(hsa_init());
    CK(hsa_iterate_agents(on_agent, nullptr));
    if (!g_have_gpu || !g_have_cpu) { fprintf(stderr, "no GPU agent\n"); return 2; }
    CK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_gpu_pool, nullptr));
    CK(hsa_amd_agent_iterate_memory_pools(g_cpu, on_cpu_pool, nullptr));

    FILE *f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> blob(sz); if (fread(blob.data(), 1, sz, f) != (size_t)sz) return 2; fclose(f);
    hsa_code_object_reader_t rd; CK(hsa_code_object_reader_create_from_memory(blob.data(), sz, &rd));
    hsa_executable_t ex; CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
    CK(hsa_executable_load_agent_code_object(ex, g_gpu, rd, nullptr, nullptr));
    CK(hsa_executable_freeze(ex, nullptr));
    const Kernel k_empty = get_kernel(ex, "k_empty.kd"), k_chain = get_kernel(ex, "k_chain.kd"), k_plain = get_kernel(ex, "k_chain_plain.kd");

    hsa_queue_t *q; CK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    hsa_signal_t done; CK(hsa_signal_create(1, 0, nullptr, &done));

    uint32_t *bufs; CK(hsa_amd_memory_pool_allocate(g_dev_pool, 2 * GRID * sizeof(uint32_t), 0, (void **)&bufs));
    uint32_t *host; CK(hsa_amd_memory_pool_allocate(g_fine_pool, 2 * GRID * sizeof(uint32_t), 0, (void **)&host));
    CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, host));
    // kernel arguments in DEVICE memory, as HIP keeps them on this part (HIP_FORCE_DEV_KERNARG); `host` kernarg pool kept for comparison
    const bool dev_kernarg = !(argc > 3 && !strcmp(argv[3], "host-kernarg"));
    ChainArgs *args, *hargs; CK(hsa_amd_memory_pool_allocate(g_kernarg_pool, 2 * sizeof(ChainArgs), 0, (void **)&hargs));
    CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, hargs));
    hargs[0] = {bufs, bufs + GRID, (uint32_t)GRID, 0}; hargs[1] = {bufs + GRID, bufs, (uint32_t)GRID, 0};
    if (dev_kernarg) { CK(hsa_amd_memory_pool_allocate(g_dev_pool, 4096, 0, (void **)&args)); CK(hsa_memory_copy(args, hargs, 2 * sizeof(ChainArgs))); }
    else args = hargs;
    printf("kernel arguments in %s memory\n", dev_kernarg ? "device" : "host (kernarg pool)");

    auto header = [](bool barrier, int acq, int rel) {
        return (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                          (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    };
    // One chain: N dispatches; the first acquires and the last releases at system scope and carries the completion signal, the ones
    // between use (barrier, acq, rel).  Returns us per dispatch (host clock around the whole chain; N is large).
    auto run_chain = [&](const Kernel &k, bool barrier, int acq, int rel, bool pingpong) {
        memset(host, 0, 2 * GRID * sizeof(uint32_t));
        CK(hsa_memory_copy(bufs, host, 2 * GRID * sizeof(uint32_t)));
        hsa_signal_store_relaxed(done, 1);
        uint64_t last_idx = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) {
            const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
            while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {}
            hsa_kernel_dispatch_packet_t *p = (hsa_kernel_dispatch_packet_t *)q->base_address + (idx & (q->size - 1));
            p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            p->workgroup_size_x = BLOCK; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
            p->grid_size_x = GRID * BLOCK; p->grid_size_y = 1; p->grid_size_z = 1;
            p->private_segment_size = k.priv; p->group_segment_size = k.group;
            p->kernel_object = k.object; p->kernarg_address = &args[pingpong ? (i & 1) : 0]; p->reserved2 = 0;
            const bool first = i == 0, last = i == N - 1;
            p->completion_signal.handle = last ? done.handle : 0;
            const uint16_t h = header(first || last ? true : barrier, first ? HSA_FENCE_SCOPE_SYSTEM : acq, last ? HSA_FENCE_SCOPE_SYSTEM : rel);
            __atomic_store_n((uint16_t *)p, h, __ATOMIC_RELEASE);
            last_idx = idx;
            if (per_packet_doorbell) hsa_signal_store_screlease(q->doorbell_signal, idx);
        }
        const auto t1 = std::chrono::steady_clock::now();
        if (!per_packet_doorbell) hsa_signal_store_screlease(q->doorbell_signal, last_idx);   // the whole chain behind ONE doorbell, like a graph replay
        while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) {}
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - (per_packet_doorbell ? t0 : t1)).count();
        return us / N;
    };
    auto check = [&]() {   // every block's counter must have advanced once per dispatch: N on the buffer the last dispatch wrote
        CK(hsa_memory_copy(host, bufs, 2 * GRID * sizeof(uint32_t)));
        const uint32_t *last = host + ((N & 1) ? GRID : 0);
        int bad = 0; uint32_t lo = UINT32_MAX;
        for (int b = 0; b < GRID; b++) { if (last[b] != (uint32_t)N) bad++; if (last[b] < lo) lo = last[b]; }
        return bad ? (printf("  STALE: %d of %d blocks behind (lowest %u of %d)", bad, GRID, lo, N), false) : (printf("  values ok"), true);
    };
    struct V { const char *name; bool barrier; int acq, rel; };
    const V vs[] = {{"barrier, acquire system / release system", true, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM},
                    {"barrier, acquire agent  / release agent ", true, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT},
                    {"barrier, acquire agent  / release none  ", true, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_NONE},
                    {"barrier, acquire none   / release agent ", true, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_AGENT},
                    {"barrier, acquire none   / release none  ", true, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE},
                    {"no barrier, no fences (independent)     ", false, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE}};
    for (int r = 0; r < 20; r++) run_chain(k_empty, true, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, false);   // clocks up
    printf("chain of %d dispatches, grid %d x %d threads, one user-mode queue (us per dispatch, best of 3)\n", N, GRID, BLOCK);
    for (const V &v : vs) {
        double e = 1e9, c = 1e9, pl = 1e9;
        for (int r = 0; r < 3; r++) { double t = run_chain(k_empty, v.barrier, v.acq, v.rel, false); if (t < e) e = t; }
        printf("%s: empty %.2f", v.name, e);
        for (int r = 0; r < 3; r++) { double t = run_chain(k_chain, v.barrier, v.acq, v.rel, true); if (t < c) c = t; }
        printf(" | chain (write-through stores, L2-bypassing loads) %.2f", c);
        if (v.barrier) check(); else printf("  (unordered: values not checked)");
        for (int r = 0; r < 3; r++) { double t = run_chain(k_plain, v.barrier, v.acq, v.rel, true); if (t < pl) pl = t; }
        printf(" | chain (plain loads / stores) %.2f", pl);
        if (v.barrier) check(); else printf("  (unordered)");
        printf("\n"); fflush(stdout);
    }
    hsa_queue_destroy(q); hsa_shut_down();
    return 0;
}
