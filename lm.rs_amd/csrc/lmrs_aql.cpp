// lmrs_aql.cpp — HSA side of the hand-written AQL decode step (see lmrs_aql.h).
#include "lmrs_aql.h"

#include <cxxabi.h>
#include <dlfcn.h>
#include <elf.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include <map>
#include <mutex>
#include <string>

namespace lmrs {

static thread_local AqlRecorder* t_rec = nullptr;
AqlRecorder* aql_recorder() { return t_rec; }
void aql_set_recorder(AqlRecorder* r) { t_rec = r; }

namespace {

std::string hsa_msg(const char* what, hsa_status_t s) {
    const char* m = nullptr;
    hsa_status_string(s, &m);
    return std::string(what) + ": " + (m ? m : "unknown HSA status");
}
#define HSA_TRY(expr)                                                           \
    do {                                                                        \
        const hsa_status_t s_ = (expr);                                         \
        if (s_ != HSA_STATUS_SUCCESS) { if (err) *err = hsa_msg(#expr, s_); return false; } \
    } while (0)

// ---- the gfx950 code objects embedded in this shared library (.hip_fatbin: one clang offload bundle per translation unit)
struct Blob { const char* p; size_t n; };
std::vector<char> g_self;                                   // the library file (kept: the HSA loader reads the code objects in place)
bool find_code_objects(std::vector<Blob>& out, std::string* err) {
    Dl_info di{};
    if (!dladdr(reinterpret_cast<const void*>(&aql_recorder), &di) || !di.dli_fname) { if (err) *err = "dladdr failed"; return false; }
    if (g_self.empty()) {
        FILE* f = fopen(di.dli_fname, "rb");
        if (!f) { if (err) *err = std::string("cannot read ") + di.dli_fname; return false; }
        fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
        g_self.resize((size_t)sz);
        const size_t got = fread(g_self.data(), 1, (size_t)sz, f);
        fclose(f);
        if (got != (size_t)sz) { g_self.clear(); if (err) *err = "short read of the library file"; return false; }
    }
    const char* base = g_self.data(); const size_t n = g_self.size();
    if (n < sizeof(Elf64_Ehdr) || memcmp(base, ELFMAG, SELFMAG) != 0) { if (err) *err = "not an ELF file"; return false; }
    const Elf64_Ehdr* eh = reinterpret_cast<const Elf64_Ehdr*>(base);
    if (eh->e_shoff == 0 || eh->e_shoff + (size_t)eh->e_shnum * sizeof(Elf64_Shdr) > n || eh->e_shstrndx >= eh->e_shnum) { if (err) *err = "no section headers"; return false; }
    const Elf64_Shdr* sh = reinterpret_cast<const Elf64_Shdr*>(base + eh->e_shoff);
    const char* names = base + sh[eh->e_shstrndx].sh_offset;
    static const char kMagic[] = "__CLANG_OFFLOAD_BUNDLE__";
    for (int i = 0; i < eh->e_shnum; ++i) {
        if (strcmp(names + sh[i].sh_name, ".hip_fatbin") != 0) continue;
        const char* s = base + sh[i].sh_offset; const size_t sn = sh[i].sh_size;
        if (sh[i].sh_offset + sn > n) continue;
        for (size_t off = 0; off + 32 <= sn;) {
            if (memcmp(s + off, kMagic, 24) != 0) { off += 8; continue; }           // bundles are 4096-aligned; be lenient
            uint64_t cnt; memcpy(&cnt, s + off + 24, 8);
            size_t q = off + 32, end = off + 32;
            for (uint64_t e = 0; e < cnt && q + 24 <= sn; ++e) {
                uint64_t eo, es, tl; memcpy(&eo, s + q, 8); memcpy(&es, s + q + 8, 8); memcpy(&tl, s + q + 16, 8);
                if (q + 24 + tl > sn) break;
                const std::string triple(s + q + 24, tl);
                q += 24 + tl;
                if (off + eo + es > sn) continue;
                if (off + eo + es > end) end = off + eo + es;
                if (es && triple.find("amdgcn") != std::string::npos && triple.find("gfx950") != std::string::npos) out.push_back({s + off + eo, (size_t)es});
            }
            off = (end + 7) & ~(size_t)7;
        }
    }
    if (out.empty()) { if (err) *err = "no gfx950 code object found in the library's .hip_fatbin section"; return false; }
    return true;
}

struct KernelInfo { uint64_t object; uint32_t kernarg, group, priv; };

struct Device {
    bool ok = false; std::string why;
    hsa_agent_t agent{};
    hsa_queue_t* queue = nullptr;
    hsa_signal_t done{};
    std::vector<hsa_executable_t> exes;
    std::map<std::string, KernelInfo> by_name;              // mangled kernel name -> descriptor
    std::map<const void*, KernelInfo> by_fn;                // host function -> descriptor (resolved on first use)
    std::mutex mu;
};
std::mutex g_mu;
std::map<int, Device*> g_dev;
bool g_hsa = false;

struct AgentPick { uint32_t want_bdf, want_domain; int want_index; int seen; hsa_agent_t by_bdf, by_index; bool have_bdf, have_index; };
hsa_status_t agent_cb(hsa_agent_t ag, void* data) {
    AgentPick* p = static_cast<AgentPick*>(data);
    hsa_device_type_t t;
    if (hsa_agent_get_info(ag, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0, domain = 0;
    if (hsa_agent_get_info(ag, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS &&
        hsa_agent_get_info(ag, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain) == HSA_STATUS_SUCCESS && bdf == p->want_bdf && domain == p->want_domain && !p->have_bdf) { p->by_bdf = ag; p->have_bdf = true; }
    if (p->seen == p->want_index) { p->by_index = ag; p->have_index = true; }
    ++p->seen;
    return HSA_STATUS_SUCCESS;
}
hsa_status_t symbol_cb(hsa_executable_t, hsa_agent_t, hsa_executable_symbol_t sym, void* data) {
    Device* d = static_cast<Device*>(data);
    hsa_symbol_kind_t kind;
    if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind) != HSA_STATUS_SUCCESS || kind != HSA_SYMBOL_KIND_KERNEL) return HSA_STATUS_SUCCESS;
    uint32_t len = 0;
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_NAME_LENGTH, &len);
    std::string name(len, '\0');
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_NAME, name.data());
    while (!name.empty() && name.back() == '\0') name.pop_back();
    if (name.size() > 3 && name.compare(name.size() - 3, 3, ".kd") == 0) name.resize(name.size() - 3);
    KernelInfo k{};
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object);
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg);
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group);
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv);
    d->by_name[name] = k;
    return HSA_STATUS_SUCCESS;
}
void queue_error_cb(hsa_status_t status, hsa_queue_t*, void*) {
    const char* m = nullptr; hsa_status_string(status, &m);
    fprintf(stderr, "lmrs: HSA queue error: %s\n", m ? m : "?");
}

bool device_init(Device* d, int device, std::string* err) {
    if (hipSetDevice(device) != hipSuccess || hipFree(nullptr) != hipSuccess) { if (err) *err = "no HIP device"; return false; }
    if (!g_hsa) { HSA_TRY(hsa_init()); g_hsa = true; }      // (reference-counted: the HIP runtime holds its own)
    int bus = 0, dev = 0, dom = 0;                              // the HSA agent of THIS HIP device: by PCI address, else by enumeration order
    (void)hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device); (void)hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, device);
    (void)hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, device);
    AgentPick pick{(uint32_t)((bus << 8) | (dev << 3)), (uint32_t)dom, device, 0, {}, {}, false, false};
    HSA_TRY(hsa_iterate_agents(agent_cb, &pick));
    if (pick.have_bdf) d->agent = pick.by_bdf; else if (pick.have_index) d->agent = pick.by_index; else { if (err) *err = "no matching HSA GPU agent"; return false; }
    std::vector<Blob> cos;
    if (!find_code_objects(cos, err)) return false;
    for (const Blob& b : cos) {
        hsa_code_object_reader_t rd; hsa_executable_t exe;
        HSA_TRY(hsa_code_object_reader_create_from_memory(b.p, b.n, &rd));
        HSA_TRY(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
        HSA_TRY(hsa_executable_load_agent_code_object(exe, d->agent, rd, nullptr, nullptr));
        HSA_TRY(hsa_executable_freeze(exe, nullptr));
        HSA_TRY(hsa_executable_iterate_agent_symbols(exe, d->agent, symbol_cb, d));
        d->exes.push_back(exe);
    }
    uint32_t qmax = 0;
    HSA_TRY(hsa_agent_get_info(d->agent, HSA_AGENT_INFO_QUEUE_MAX_SIZE, &qmax));
    uint32_t qsize = 16384; while (qsize > qmax) qsize >>= 1;
    HSA_TRY(hsa_queue_create(d->agent, qsize, HSA_QUEUE_TYPE_SINGLE, queue_error_cb, nullptr, UINT32_MAX, UINT32_MAX, &d->queue));
    HSA_TRY(hsa_signal_create(1, 0, nullptr, &d->done));
    return true;
}

Device* get_device(int device, std::string* err) {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_dev.find(device);
    if (it == g_dev.end()) {
        Device* d = new Device();
        std::string why;
        d->ok = device_init(d, device, &why);
        d->why = why;
        it = g_dev.emplace(device, d).first;
    }
    if (!it->second->ok) { if (err) *err = it->second->why; return nullptr; }
    return it->second;
}

bool resolve(Device* d, const void* fn, KernelInfo* out, std::string* err) {
    auto it = d->by_fn.find(fn);
    if (it != d->by_fn.end()) { *out = it->second; return true; }
    const char* nm = hipKernelNameRefByPtr(fn, nullptr);
    if (!nm) {
        hipFunction_t hf = nullptr;
        if (hipGetFuncBySymbol(&hf, fn) == hipSuccess && hf) nm = hipKernelNameRef(hf);
    }
    if (!nm) { if (err) *err = "the HIP runtime does not know the kernel's name"; return false; }
    std::string name(nm);
    if (name.size() > 3 && name.compare(name.size() - 3, 3, ".kd") == 0) name.resize(name.size() - 3);
    auto kn = d->by_name.find(name);
    if (kn == d->by_name.end()) {                           // a demangled name: compare with the demangled symbols
        for (auto& kv : d->by_name) {
            int st = 0; char* dm = abi::__cxa_demangle(kv.first.c_str(), nullptr, nullptr, &st);
            const bool same = st == 0 && dm && name == dm;
            free(dm);
            if (same) { kn = d->by_name.find(kv.first); break; }
        }
    }
    if (kn == d->by_name.end()) { if (err) *err = "kernel " + name + " is not in the loaded code objects"; return false; }
    d->by_fn[fn] = kn->second; *out = kn->second;
    return true;
}

double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

}  // namespace

struct AqlProgram {
    int device = 0;
    struct Pkt { uint64_t object; void* kernarg; uint32_t grid[3]; uint16_t block; uint32_t group, priv; };
    std::vector<Pkt> pkts;
    char* kargs = nullptr;                                   // device memory (or pinned host memory): one block per packet
    bool host_kargs = false;
};

AqlProgram* aql_program_create(int device, const AqlRecorder& rec, std::string* err) {
    Device* d = get_device(device, err);
    if (!d) return nullptr;
    std::lock_guard<std::mutex> lock(d->mu);
    if (rec.nodes.empty()) { if (err) *err = "empty record"; return nullptr; }
    std::vector<KernelInfo> ki(rec.nodes.size());
    std::vector<size_t> koff(rec.nodes.size());
    size_t total = 0;
    for (size_t i = 0; i < rec.nodes.size(); ++i) {
        const AqlNode& n = rec.nodes[i];
        if (!resolve(d, n.fn, &ki[i], err)) return nullptr;
        if (ki[i].priv != 0) { if (err) *err = "kernel uses scratch memory"; return nullptr; }
        const size_t hidden = (n.args.size() + 7) & ~(size_t)7;
        // the packed block must be what the code object expects: explicit arguments [+ the 256-byte hidden block of code-object v5]
        if (ki[i].kernarg != hidden + 256 && !(ki[i].kernarg >= n.args.size() && ki[i].kernarg <= hidden)) {
            if (err) *err = "kernel-argument block of " + std::to_string(n.args.size()) + " bytes packed for a kernel whose segment is " + std::to_string(ki[i].kernarg) + " bytes";
            return nullptr;
        }
        koff[i] = total;
        total += ((size_t)(ki[i].kernarg > hidden + 256 ? ki[i].kernarg : hidden + 256) + 255) & ~(size_t)255;
    }
    std::vector<char> host(total, 0);
    for (size_t i = 0; i < rec.nodes.size(); ++i) {
        const AqlNode& n = rec.nodes[i];
        char* k = host.data() + koff[i];
        memcpy(k, n.args.data(), n.args.size());
        // hidden arguments of code-object v5 (the block follows the explicit arguments, 8-byte aligned)
        char* h = k + ((n.args.size() + 7) & ~(size_t)7);
        const uint32_t bc[3] = {n.grid[0], n.grid[1], n.grid[2]};
        const uint16_t gs[3] = {(uint16_t)n.block, 1, 1}, rem[3] = {0, 0, 0};
        memcpy(h + 0, bc, 12); memcpy(h + 12, gs, 6); memcpy(h + 18, rem, 6);
        const uint16_t dims = n.grid[2] > 1 ? 3 : (n.grid[1] > 1 ? 2 : 1);
        memcpy(h + 64, &dims, 2);
        const uint32_t dyn = n.lds;
        memcpy(h + 120, &dyn, 4);
    }
    AqlProgram* p = new AqlProgram();
    p->device = device;
    // LMRS_AQL_HOST_KERNARG=1: the argument blocks in pinned host memory instead of device memory - rocprofv3's queue interceptor copies
    // every packet's argument block on the HOST and segfaults on a device address (the profiling recipe sets it; a few percent slower)
    p->host_kargs = getenv("LMRS_AQL_HOST_KERNARG") && atoi(getenv("LMRS_AQL_HOST_KERNARG")) != 0;
    if (hipSetDevice(device) != hipSuccess ||
        (p->host_kargs ? hipHostMalloc(reinterpret_cast<void**>(&p->kargs), total, hipHostMallocDefault) : hipMalloc(reinterpret_cast<void**>(&p->kargs), total)) != hipSuccess ||
        hipMemcpy(p->kargs, host.data(), total, p->host_kargs ? hipMemcpyHostToHost : hipMemcpyHostToDevice) != hipSuccess) {
        if (err) *err = "hipMalloc / hipMemcpy of the argument blocks failed";
        aql_program_destroy(p);
        return nullptr;
    }
    for (size_t i = 0; i < rec.nodes.size(); ++i) {
        const AqlNode& n = rec.nodes[i];
        p->pkts.push_back({ki[i].object, p->kargs + koff[i], {n.grid[0] * n.block, n.grid[1], n.grid[2]}, (uint16_t)n.block, ki[i].group + n.lds, 0});
    }
    return p;
}

void aql_program_destroy(AqlProgram* p) {
    if (!p) return;
    if (p->kargs) (void)(p->host_kargs ? hipHostFree(p->kargs) : hipFree(p->kargs));
    delete p;
}
int aql_program_launches(const AqlProgram* p) { return p ? (int)p->pkts.size() : 0; }

int aql_run(int device, AqlProgram* const* steps, size_t n, int fence_scope, double* seconds, std::string* err) {
    Device* d = get_device(device, err);
    if (!d) return -1;
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lock(d->mu);
    hsa_queue_t* q = d->queue;
    size_t total = 0;
    for (size_t s = 0; s < n; ++s) total += steps[s]->pkts.size();
    hsa_signal_store_relaxed(d->done, 1);
    const uint32_t scope = fence_scope ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE;
    hsa_kernel_dispatch_packet_t* ring = static_cast<hsa_kernel_dispatch_packet_t*>(q->base_address);
    const double t0 = now_s();
    size_t written = 0;
    for (size_t s = 0; s < n; ++s) {
        const auto& pk = steps[s]->pkts;
        const uint64_t cnt = pk.size(), idx = hsa_queue_add_write_index_relaxed(q, cnt);
        // room in the ring (the packet processor is tens of steps behind the host at most)
        for (double tw = now_s(); idx + cnt - hsa_queue_load_read_index_scacquire(q) > q->size;)
            if (now_s() - tw > 20.0) { if (err) *err = "AQL queue stalled"; return -1; }
        for (uint64_t i = 0; i < cnt; ++i) {
            hsa_kernel_dispatch_packet_t* p = ring + ((idx + i) & (q->size - 1));
            const AqlProgram::Pkt& k = pk[i];
            p->workgroup_size_x = k.block; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
            p->grid_size_x = k.grid[0]; p->grid_size_y = k.grid[1]; p->grid_size_z = k.grid[2];
            p->private_segment_size = k.priv; p->group_segment_size = k.group;
            p->kernel_object = k.object; p->kernarg_address = k.kernarg; p->reserved2 = 0;
            const bool first = written == 0, last = written + 1 == total;
            p->completion_signal = last ? d->done : hsa_signal_t{0};
            const uint32_t acq = first ? HSA_FENCE_SCOPE_SYSTEM : scope, rel = last ? HSA_FENCE_SCOPE_SYSTEM : scope;
            const uint32_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1u << HSA_PACKET_HEADER_BARRIER) |
                                    (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
            const uint32_t dims = k.grid[2] > 1 ? 3 : (k.grid[1] > 1 ? 2 : 1);
            __atomic_store_n(&p->full_header, header | (dims << 16), __ATOMIC_RELEASE);
            ++written;
        }
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(idx + cnt - 1));
    }
    for (int tries = 0;; ++tries) {
        if (hsa_signal_wait_scacquire(d->done, HSA_SIGNAL_CONDITION_LT, 1, 1000000000ull, HSA_WAIT_STATE_ACTIVE) < 1) break;
        if (tries >= 30) { if (err) *err = "AQL step did not complete within 30 s"; return -1; }
    }
    if (seconds) *seconds = now_s() - t0;
    return 0;
}

}  // namespace lmrs
