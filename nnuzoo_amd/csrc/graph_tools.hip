// hipGraph rewriting pass: every memset node of a captured graph becomes a fill-KERNEL node.
//
// Why (measured in round 1 on ROCm 7.2 / MI355X, tools/probes/graph_memset_repro.py): a captured hipMemsetAsync node
// replays with a corrupted fill pattern once the process has made further allocations - the zoo steps' replayed
// gradients came out NaN.  This library never emits hipMemsetAsync (common.hpp zero_async), but library code captured
// alongside it does: ATen's multi-block reductions zero their semaphores with it, and so do some MIOpen / hipBLASLt paths.
// Instead of avoiding those ops one by one, the captured graph is edited before instantiation:
//   for each node of type hipGraphNodeTypeMemset: read its parameters, add a kernel node (nnz::graph_fill_kernel) with the
//   same dependencies and dependents that writes the same bytes, and destroy the memset node.
// The replayed graph then contains kernel nodes only.  Host code: nnuzoo_amd/training/graph_step.py
// (torch.cuda.CUDAGraph(keep_graph=True) -> raw_cuda_graph() -> this pass -> instantiate()).
#include "common.hpp"

#include <vector>

namespace nnz {

// dst[row * pitch + i * elem .. ) = value (low `elem` bytes), i < width, row < height; elem in {1, 2, 4}
__global__ __launch_bounds__(256) void graph_fill_kernel(unsigned char* dst, size_t pitch, unsigned int value,
                                                         unsigned int elem, size_t width, size_t height) {
  const size_t row_bytes = width * elem;
  const size_t total = row_bytes * height;
  const size_t stride = (size_t)gridDim.x * 256;
  if (height == 1 && ((uintptr_t)dst & 3) == 0 && (row_bytes & 3) == 0) {
    // contiguous, word aligned: 4-byte stores of the replicated pattern
    unsigned int v = value;
    if (elem == 1) { v &= 0xFF; v |= v << 8; v |= v << 16; }
    else if (elem == 2) { v &= 0xFFFF; v |= v << 16; }
    unsigned int* d4 = reinterpret_cast<unsigned int*>(dst);
    const size_t n4 = total >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) d4[i] = v;
    return;
  }
  for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < total; b += stride) {
    const size_t row = b / row_bytes, off = b % row_bytes;
    dst[row * pitch + off] = (unsigned char)(value >> (8 * (off % elem)));
  }
}

}  // namespace nnz

// Replaces the memset nodes of `graph` (a hipGraph_t that is not instantiated yet, or will be re-instantiated) by kernel
// nodes.  *n_replaced receives the number of nodes rewritten.  Returns 0, a hipError_t, or -22.
extern "C" int nnz_graph_replace_memsets(void* graph_v, int* n_replaced) {
  using namespace nnz;
  if (!graph_v) return NNZ_EINVAL;
  hipGraph_t graph = (hipGraph_t)graph_v;
  if (n_replaced) *n_replaced = 0;
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
  if (e != hipSuccess) return (int)e;
  if (n == 0) return NNZ_OK;
  std::vector<hipGraphNode_t> nodes(n);
  e = hipGraphGetNodes(graph, nodes.data(), &n);
  if (e != hipSuccess) return (int)e;
  int replaced = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType ty;
    e = hipGraphNodeGetType(nodes[i], &ty);
    if (e != hipSuccess) return (int)e;
    if (ty != hipGraphNodeTypeMemset) continue;
    hipMemsetParams mp;
    e = hipGraphMemsetNodeGetParams(nodes[i], &mp);
    if (e != hipSuccess) return (int)e;
    if (mp.elementSize != 1 && mp.elementSize != 2 && mp.elementSize != 4) return NNZ_EINVAL;
    size_t ndep = 0, nout = 0;
    e = hipGraphNodeGetDependencies(nodes[i], nullptr, &ndep);
    if (e != hipSuccess) return (int)e;
    std::vector<hipGraphNode_t> deps(ndep ? ndep : 1);
    if (ndep) {
      e = hipGraphNodeGetDependencies(nodes[i], deps.data(), &ndep);
      if (e != hipSuccess) return (int)e;
    }
    e = hipGraphNodeGetDependentNodes(nodes[i], nullptr, &nout);
    if (e != hipSuccess) return (int)e;
    std::vector<hipGraphNode_t> outs(nout ? nout : 1);
    if (nout) {
      e = hipGraphNodeGetDependentNodes(nodes[i], outs.data(), &nout);
      if (e != hipSuccess) return (int)e;
    }
    unsigned char* dst = (unsigned char*)mp.dst;
    size_t pitch = mp.pitch, width = mp.width, height = mp.height ? mp.height : 1;
    unsigned int value = mp.value, elem = mp.elementSize;
    void* args[] = {&dst, &pitch, &value, &elem, &width, &height};
    const size_t bytes = width * elem * height;
    size_t blocks = (bytes / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipKernelNodeParams kp = {};
    kp.func = reinterpret_cast<void*>(graph_fill_kernel);
    kp.gridDim = dim3((unsigned)blocks);
    kp.blockDim = dim3(256);
    kp.sharedMemBytes = 0;
    kp.kernelParams = args;
    kp.extra = nullptr;
    hipGraphNode_t knode;
    e = hipGraphAddKernelNode(&knode, graph, ndep ? deps.data() : nullptr, ndep, &kp);
    if (e != hipSuccess) return (int)e;
    for (size_t j = 0; j < nout; ++j) {
      e = hipGraphAddDependencies(graph, &knode, &outs[j], 1);
      if (e != hipSuccess) return (int)e;
    }
    e = hipGraphDestroyNode(nodes[i]);
    if (e != hipSuccess) return (int)e;
    ++replaced;
  }
  if (n_replaced) *n_replaced = replaced;
  return NNZ_OK;
}

// Census of a graph's node types (diagnostics): counts[t] += 1 for hipGraphNodeType t < ncounts.
extern "C" int nnz_graph_node_census(void* graph_v, int* counts, int ncounts) {
  if (!graph_v || !counts || ncounts < 1) return NNZ_EINVAL;
  hipGraph_t graph = (hipGraph_t)graph_v;
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
  if (e != hipSuccess) return (int)e;
  std::vector<hipGraphNode_t> nodes(n ? n : 1);
  if (n) {
    e = hipGraphGetNodes(graph, nodes.data(), &n);
    if (e != hipSuccess) return (int)e;
  }
  for (int t = 0; t < ncounts; ++t) counts[t] = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType ty;
    e = hipGraphNodeGetType(nodes[i], &ty);
    if (e != hipSuccess) return (int)e;
    if ((int)ty >= 0 && (int)ty < ncounts) counts[(int)ty] += 1;
  }
  return NNZ_OK;
}
