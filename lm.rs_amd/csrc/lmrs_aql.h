// lmrs_aql.h — the decode step as hand-written AQL packets on an HSA user-mode queue.
//
// Why: a decode step is 49-65 short dependent kernels and a quarter of its time is kernel boundaries.  Measured on MI355X
// (tools/ubench/aql.hip, a chain of 400 trivial dependent kernels): 1.97 us per kernel replayed from a hipGraph, 2.51 us as eager HIP
// launches, 1.64 us as AQL dispatch packets written straight into an HSA queue with the same agent-scope fences HIP uses, 1.39 us
// with no fences.  (A packet WITHOUT the barrier bit does not start earlier on this chip - it waits until the packet before it has
// all but drained: no overlap of dependent launches to be had that way, which is also why HIP ignores hipExtAnyOrderLaunch on gfx9.)
//
// How: the launches of one step are RECORDED instead of enqueued (LMRS_LAUNCH_GRID consults aql_recorder()): kernel, grid, LDS bytes
// and the argument block, laid out as the code object's metadata prescribes (explicit arguments at their natural alignment, then the
// hidden block of code-object v5).  The record is turned into a program: kernel objects resolved through the HSA loader from the
// code objects embedded in this very library (the same bytes the HIP runtime loads), argument blocks uploaded once to device memory -
// position and tokens live on the device, so a step needs no per-step arguments.  lmrs_generate_greedy then writes the packets of as
// many steps as it has to run, rings the doorbell once and waits for the last packet's completion signal.
// Memory (weights, KV cache, activations) stays with HIP; HIP stream and HSA queue are ordered through the host (one synchronise
// before the first packet; the first packet acquires, the last releases, at system scope).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>

namespace lmrs {

struct AqlNode {
    const void* fn;                 // host-side kernel function (the name the loader knows it by is looked up from this)
    unsigned grid[3], block, lds;   // workgroups per dimension, threads per workgroup, dynamic LDS bytes
    std::vector<char> args;         // explicit kernel arguments, packed
};
struct AqlRecorder { std::vector<AqlNode> nodes; };

// thread-local: non-null while a step is being recorded instead of launched
AqlRecorder* aql_recorder();
void aql_set_recorder(AqlRecorder* r);

template <class T> inline void aql_pack(std::vector<char>& buf, const T& v) {
    const size_t off = (buf.size() + alignof(T) - 1) & ~(alignof(T) - 1);
    buf.resize(off + sizeof(T));
    memcpy(buf.data() + off, &v, sizeof(T));
}
template <class... A> inline void aql_record(const void* fn, dim3 grid, unsigned nt, size_t smem, const A&... a) {
    AqlNode n{fn, {grid.x, grid.y, grid.z}, nt, (unsigned)smem, {}};
    (aql_pack(n.args, a), ...);
    aql_recorder()->nodes.push_back(std::move(n));
}

struct AqlProgram;                  // the packets of one step (kernel objects + device-resident argument blocks)
// build: null + message in *err when any kernel cannot be resolved (the caller keeps the hipGraph path)
AqlProgram* aql_program_create(int device, const AqlRecorder& rec, std::string* err);
void aql_program_destroy(AqlProgram* p);
int aql_program_launches(const AqlProgram* p);
// run steps[0..n) back to back on the device's queue and wait; fence_scope: 1 agent (what HIP does between kernels), 0 none.
// -> 0, or -1 with a message in *err (timeout, queue error)
int aql_run(int device, AqlProgram* const* steps, size_t n, int fence_scope, double* seconds, std::string* err);

}  // namespace lmrs
