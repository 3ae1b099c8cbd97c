// aql.hip — what does a dependent dispatch cost when the AQL packets are written by hand, and what does a dispatch WITHOUT the
// barrier bit buy?  (26 % of a decode layer is kernel boundaries; HIP sets the barrier bit and agent-scope fences on
// every in-stream launch and ignores hipExtAnyOrderLaunch on gfx9.)
//
// Memory comes from HIP (hipMalloc), the queue, the code object and the packets from the HSA runtime underneath it.
//   T1  chain of N dependent trivial kernels: barrier bit set, acquire / release fence scope none | agent | system; against a hipGraph
//       of the same chain
//   T2  the same chain with the barrier bit clear (dispatch rate)
//   T3  A (256 workgroups spinning 20 us) then B with the barrier bit clear: when does B start?
//   T4  A with 8192 short workgroups (four residency rounds) then B, barrier bit clear: does B's first workgroup start before A's
//       last one has started (is the dispatch order across packets of one queue strict)?
//   T5  hand-off: A's workgroups publish {value, tag} granules, B (barrier bit clear) polls them - last store -> all seen,
//       against the same pair with the barrier bit set
// build: see Makefile (aql: the kernels are compiled twice, into the host binary for the hipGraph leg and into aql_kernels.hsaco)
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <vector>

struct KArgs { unsigned long long* stamps; unsigned long long* gran; int spin; unsigned tag; int nprod; int pad; };

extern "C" __global__ void k_chain(KArgs a) {
    if (threadIdx.x == 0 && blockIdx.x == 0) a.stamps[0] = wall_clock64();
    unsigned long long* p = a.gran + blockIdx.x * 256 + threadIdx.x;
    const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p, v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0 && blockIdx.x == 0) a.stamps[1] = wall_clock64();
}
extern "C" __global__ void k_spin(KArgs a) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)a.spin) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 0) { a.stamps[2 * blockIdx.x] = t0; a.stamps[2 * blockIdx.x + 1] = wall_clock64(); }
}
extern "C" __global__ void k_probe(KArgs a) {
    if (threadIdx.x == 0) { a.stamps[2 * blockIdx.x] = wall_clock64(); a.stamps[2 * blockIdx.x + 1] = wall_clock64(); }
}
extern "C" __global__ void k_produce(KArgs a) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)a.spin) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 0) {
        __hip_atomic_store(a.gran + blockIdx.x, ((unsigned long long)a.tag << 32) | blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.stamps[2 * blockIdx.x] = t0; a.stamps[2 * blockIdx.x + 1] = wall_clock64();
    }
}
extern "C" __global__ void k_consume(KArgs a) {
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    for (;; ++spins) {
        int ok = 1;
        for (int i = threadIdx.x; i < a.nprod; i += 256) {
            const unsigned long long g = __hip_atomic_load(a.gran + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(g >> 32) != a.tag) ok = 0;
        }
        if (__syncthreads_and(ok) || spins > (1u << 16)) break;
    }
    if (threadIdx.x == 0) { a.stamps[2 * blockIdx.x] = t0; a.stamps[2 * blockIdx.x + 1] = wall_clock64() | (spins > (1u << 16) ? 1ull << 63 : 0); }
}

#ifndef __HIP_DEVICE_COMPILE__
#define HSA(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = ""; hsa_status_string(s_, &m); fprintf(stderr, "%s: %s\n", #x, m); exit(2); } } while (0)
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static hsa_agent_t g_gpu; static bool g_have = false;
static hsa_status_t pick_gpu(hsa_agent_t ag, void*) {
    hsa_device_type_t t; hsa_agent_get_info(ag, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have) { g_gpu = ag; g_have = true; }
    return HSA_STATUS_SUCCESS;
}
struct Kern { uint64_t object; uint32_t kernarg, group, priv; };
static Kern get_kernel(hsa_executable_t exe, const char* name) {
    hsa_executable_symbol_t sym; char kd[128]; snprintf(kd, sizeof kd, "%s.kd", name);
    HSA(hsa_executable_get_symbol_by_name(exe, kd, &g_gpu, &sym));
    Kern k{};
    HSA(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    HSA(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
    HSA(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    HSA(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    return k;
}
static hsa_queue_t* g_q;
struct Disp { Kern k; unsigned grid; void* kernarg; int barrier, acq, rel; hsa_signal_t sig; };
static void submit(const std::vector<Disp>& d) {
    const uint64_t n = d.size(), idx = hsa_queue_add_write_index_relaxed(g_q, n);
    while (idx + n - hsa_queue_load_read_index_scacquire(g_q) > g_q->size) {}
    hsa_kernel_dispatch_packet_t* base = (hsa_kernel_dispatch_packet_t*)g_q->base_address;
    for (uint64_t i = 0; i < n; ++i) {
        hsa_kernel_dispatch_packet_t* p = base + ((idx + i) & (g_q->size - 1));
        p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
        p->grid_size_x = d[i].grid * 256; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = d[i].k.priv; p->group_segment_size = d[i].k.group;
        p->kernel_object = d[i].k.object; p->kernarg_address = d[i].kernarg; p->reserved2 = 0; p->completion_signal = d[i].sig;
        const uint32_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((uint32_t)d[i].barrier << HSA_PACKET_HEADER_BARRIER) |
                                ((uint32_t)d[i].acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | ((uint32_t)d[i].rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        __atomic_store_n(&p->full_header, header | (1u << 16), __ATOMIC_RELEASE);          // setup: one dimension
    }
    hsa_signal_store_screlease(g_q->doorbell_signal, (hsa_signal_value_t)(idx + n - 1));
}
static bool wait_sig(hsa_signal_t s) {
    for (int i = 0; i < 50; ++i)
        if (hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, 200000000ull, HSA_WAIT_STATE_ACTIVE) < 1) return true;
    return false;
}
static double now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

int main(int argc, char** argv) {
    const char* hsaco = argc > 1 ? argv[1] : "tools/ubench/aql_kernels.hsaco";
    HIPC(hipSetDevice(0));
    HIPC(hipFree(nullptr));
    HSA(hsa_init());
    HSA(hsa_iterate_agents(pick_gpu, nullptr));
    if (!g_have) { fprintf(stderr, "no GPU agent\n"); return 2; }
    FILE* f = fopen(hsaco, "rb"); if (!f) { perror(hsaco); return 2; }
    fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> img(sz); if (fread(img.data(), 1, sz, f) != (size_t)sz) return 2; fclose(f);
    hsa_code_object_reader_t rd; hsa_executable_t exe;
    HSA(hsa_code_object_reader_create_from_memory(img.data(), sz, &rd));
    HSA(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HSA(hsa_executable_load_agent_code_object(exe, g_gpu, rd, nullptr, nullptr));
    HSA(hsa_executable_freeze(exe, nullptr));
    const Kern kc = get_kernel(exe, "k_chain"), ks = get_kernel(exe, "k_spin"), kp = get_kernel(exe, "k_probe"), kpr = get_kernel(exe, "k_produce"), kco = get_kernel(exe, "k_consume");
    printf("kernel objects loaded: k_chain kernarg %u B, group %u, private %u\n", kc.kernarg, kc.group, kc.priv);
    HSA(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &g_q));
    hsa_signal_t sig; HSA(hsa_signal_create(1, 0, nullptr, &sig));
    const hsa_signal_t nosig{0};

    const int NS = 2 * 8192;
    unsigned long long *stA, *stB, *gran; char* kargs;
    HIPC(hipMalloc(&stA, NS * 8)); HIPC(hipMalloc(&stB, NS * 8)); HIPC(hipMalloc(&gran, 256 * 256 * 8)); HIPC(hipMalloc(&kargs, 4096 * 64));
    HIPC(hipMemset(gran, 0, 256 * 256 * 8));
    std::vector<unsigned long long> hA(NS), hB(NS);
    auto set_args = [&](int slot, const KArgs& a) { HIPC(hipMemcpy(kargs + slot * 64, &a, sizeof a, hipMemcpyHostToDevice)); return (void*)(kargs + slot * 64); };

    // ---- T1 / T2: chain of N trivial dependent kernels
    const int N = 400;
    void* ka_chain = set_args(0, KArgs{stA, gran, 0, 0, 0, 0});
    HIPC(hipDeviceSynchronize());
    for (int barrier = 1; barrier >= 0; --barrier)
        for (int scope = 0; scope <= 2; ++scope) {
            double best = 1e30;
            for (int rep = 0; rep < 5; ++rep) {
                std::vector<Disp> d(N, Disp{kc, 256, ka_chain, barrier, scope, scope, nosig});
                d[0].barrier = 1; d[0].acq = 2; d[N - 1].rel = 2; d[N - 1].barrier = 1; d[N - 1].sig = sig;
                hsa_signal_store_relaxed(sig, 1);
                const double t0 = now_us();
                submit(d);
                if (!wait_sig(sig)) { fprintf(stderr, "T1 timeout\n"); return 3; }
                best = std::min(best, now_us() - t0);
            }
            printf("T%d chain of %d trivial kernels (256 x 256), barrier bit %d, fences %s: %.3f us per kernel (host wall, best of 5)\n", barrier ? 1 : 2, N, barrier,
                   scope == 0 ? "none" : scope == 1 ? "agent" : "system", best / N);
        }
    {   // the same chain as a hipGraph
        hipStream_t s; HIPC(hipStreamCreate(&s));
        hipGraph_t g; hipGraphExec_t ge;
        HIPC(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, s, KArgs{stA, gran, 0, 0, 0, 0});
        HIPC(hipStreamEndCapture(s, &g)); HIPC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double best = 1e30;
        for (int rep = 0; rep < 6; ++rep) { HIPC(hipStreamSynchronize(s)); const double t0 = now_us(); HIPC(hipGraphLaunch(ge, s)); HIPC(hipStreamSynchronize(s)); if (rep) best = std::min(best, now_us() - t0); }
        printf("T1 the same chain as a hipGraph: %.3f us per kernel\n", best / N);
        best = 1e30;
        for (int rep = 0; rep < 6; ++rep) { HIPC(hipStreamSynchronize(s)); const double t0 = now_us(); for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, s, KArgs{stA, gran, 0, 0, 0, 0}); HIPC(hipStreamSynchronize(s)); if (rep) best = std::min(best, now_us() - t0); }
        printf("T1 the same chain as eager HIP launches: %.3f us per kernel\n", best / N);
    }

    auto run_pair = [&](const Kern& ka, unsigned ga, const KArgs& aa, const Kern& kb, unsigned gb, const KArgs& ab, int barrier_b, int scope) {
        void* pa = set_args(1, aa); void* pb = set_args(2, ab);
        HIPC(hipMemset(stA, 0, NS * 8)); HIPC(hipMemset(stB, 0, NS * 8)); HIPC(hipDeviceSynchronize());
        std::vector<Disp> d{Disp{ka, ga, pa, 1, 2, scope, nosig}, Disp{kb, gb, pb, barrier_b, scope, 2, sig}};
        hsa_signal_store_relaxed(sig, 1);
        submit(d);
        if (!wait_sig(sig)) { fprintf(stderr, "pair timeout\n"); exit(3); }
        HIPC(hipMemcpy(hA.data(), stA, NS * 8, hipMemcpyDeviceToHost)); HIPC(hipMemcpy(hB.data(), stB, NS * 8, hipMemcpyDeviceToHost));
    };
    auto mm = [&](const std::vector<unsigned long long>& h, unsigned n, int which, bool mx) {
        unsigned long long r = mx ? 0 : ~0ull;
        for (unsigned i = 0; i < n; ++i) { const unsigned long long v = h[2 * i + which] & ~(1ull << 63); r = mx ? std::max(r, v) : std::min(r, v); }
        return r;
    };
    // ---- T3
    for (int barrier = 0; barrier <= 1; ++barrier) {
        run_pair(ks, 256, KArgs{stA, gran, 2000, 0, 0, 0}, kp, 256, KArgs{stB, gran, 0, 0, 0, 0}, barrier, 1);
        const unsigned long long a0 = mm(hA, 256, 0, false);
        printf("T3 A = 256 workgroups spinning 20 us, B barrier bit %d: A first start 0, A last end %.2f us, B first start %.2f us, B last start %.2f us\n", barrier,
               (mm(hA, 256, 1, true) - a0) * 0.01, ((double)mm(hB, 256, 0, false) - (double)a0) * 0.01, ((double)mm(hB, 256, 0, true) - (double)a0) * 0.01);
    }
    // ---- T4
    for (int rep = 0; rep < 3; ++rep) {
        run_pair(ks, 8192, KArgs{stA, gran, 300, 0, 0, 0}, kp, 256, KArgs{stB, gran, 0, 0, 0, 0}, 0, 1);
        const unsigned long long a0 = mm(hA, 8192, 0, false);
        printf("T4 A = 8192 workgroups of 3 us, B barrier bit 0: A last START %.2f us, A last end %.2f us, B first start %.2f us, B last start %.2f us\n",
               (mm(hA, 8192, 0, true) - a0) * 0.01, (mm(hA, 8192, 1, true) - a0) * 0.01, ((double)mm(hB, 256, 0, false) - (double)a0) * 0.01, ((double)mm(hB, 256, 0, true) - (double)a0) * 0.01);
    }
    // ---- T5
    for (int scope = 0; scope <= 1; ++scope)
        for (int barrier = 0; barrier <= 1; ++barrier)
            for (int rep = 0; rep < 3; ++rep) {
                const unsigned tag = 100 + scope * 50 + barrier * 10 + rep;
                run_pair(kpr, 256, KArgs{stA, gran, 500, tag, 0, 0}, kco, 256, KArgs{stB, gran, 0, tag, 256, 0}, barrier, scope);
                const unsigned long long a1 = mm(hA, 256, 1, true);
                bool to = false; for (int i = 0; i < 256; ++i) to |= (hB[2 * i + 1] >> 63) != 0;
                printf("T5 hand-off of 256 granules to 256 workgroups, barrier bit %d, fences %s: last publish -> B first start %.2f us, -> first all-seen %.2f us, -> last all-seen %.2f us%s\n", barrier,
                       scope ? "agent" : "none", ((double)mm(hB, 256, 0, false) - (double)a1) * 0.01, ((double)mm(hB, 256, 1, false) - (double)a1) * 0.01, ((double)mm(hB, 256, 1, true) - (double)a1) * 0.01, to ? "  (POLL TIMEOUT)" : "");
            }
    hsa_queue_destroy(g_q);
    printf("done\n");
    return 0;
}
#endif
