// srt_lbvh.hip -- device-side BVH build: a linear BVH (Morton order + Karras'
// parallel hierarchy + bottom-up refit) over the primitives already resident in HBM.
//
// This is NOT the reference's tree: bvh.h:55-95 builds a random-axis median split whose shape the
// FAITHFUL traversal semantics depend on.  A device-built tree is for
// SRT_TRAVERSE_CLOSEST rendering of large scenes, where the closest hit does not depend on the tree
// (the reference's tree needs 2 400+ node visits per ray on a 10 M-triangle soup).
// Primitive boxes follow the reference's boundingBox rules (model.h:183-212 incl. the +-1e-4 padding
// of flat axes; sphere.h:85-94 incl. motion over [time0, time1]).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <utility>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "srt_device.h"

namespace {

__device__ __forceinline__ int orderedInt(float f) {  // monotone float -> int for atomicMin/Max
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ __forceinline__ float orderedFloat(int i) {
  int b = i >= 0 ? i : i ^ 0x7fffffff;
  float f;
  memcpy(&f, &b, 4);
  return f;
}

__device__ __forceinline__ void primBox(const DevScene& sc, int ref, float time0, float time1, float* mn, float* mx) {
  int pr = ~ref;
  if (pr & 1) {  // sphere.h:85-94
    const float4* sp = sc.spheres + 3 * (pr >> 1);
    float4 s0 = sp[0], s1 = sp[1], s2 = sp[2];
    float c0[3] = {s0.x, s0.y, s0.z}, c1[3] = {s1.x, s1.y, s1.z};
    bool moving = __float_as_int(s1.w) & (1 << 30);
    for (int k = 0; k < 3; ++k) {
      float a = c0[k], b = c0[k];
      if (moving) {
        a = c0[k] + ((time0 - s2.x) / (s2.y - s2.x)) * (c1[k] - c0[k]);
        b = c0[k] + ((time1 - s2.x) / (s2.y - s2.x)) * (c1[k] - c0[k]);
      }
      mn[k] = fminf(a - s0.w, b - s0.w);
      mx[k] = fmaxf(a + s0.w, b + s0.w);
    }
  } else {  // model.h:183-212
    const float4* tr = sc.triTest + 3 * (pr >> 1);
    float4 q0 = tr[0], q1 = tr[1], q2 = tr[2];
    float v[3][3] = {{q0.x, q0.y, q0.z}, {q1.x, q1.y, q1.z}, {q2.x, q2.y, q2.z}};
    for (int k = 0; k < 3; ++k) {
      mn[k] = fminf(v[0][k], fminf(v[1][k], v[2][k]));
      mx[k] = fmaxf(v[0][k], fmaxf(v[1][k], v[2][k]));
      if (mn[k] == mx[k]) {
        mn[k] -= 0.0001f;
        mx[k] += 0.0001f;
      }
    }
  }
}

// 1. per-primitive boxes + scene bounds of the centroids
__global__ void lbvhPrimBoxes(DevScene sc, const int32_t* refs, int n, float time0, float time1, float4* boxMin,
                              float4* boxMax, int* bounds /* 3 min, 3 max as ordered ints */) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float mn[3], mx[3];
  primBox(sc, refs[i], time0, time1, mn, mx);
  boxMin[i] = make_float4(mn[0], mn[1], mn[2], 0.0f);
  boxMax[i] = make_float4(mx[0], mx[1], mx[2], 0.0f);
  for (int k = 0; k < 3; ++k) {
    float c = 0.5f * (mn[k] + mx[k]);
    atomicMin(&bounds[k], orderedInt(c));
    atomicMax(&bounds[3 + k], orderedInt(c));
  }
}

__device__ __forceinline__ uint32_t expandBits(uint32_t v) {  // 10 bits -> every third bit
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

// 2. 30-bit Morton code of the centroid, made unique by the primitive's position
__global__ void lbvhMorton(const float4* boxMin, const float4* boxMax, int n, const int* bounds, unsigned long long* keys,
                           int* vals) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float lo[3] = {orderedFloat(bounds[0]), orderedFloat(bounds[1]), orderedFloat(bounds[2])};
  float hi[3] = {orderedFloat(bounds[3]), orderedFloat(bounds[4]), orderedFloat(bounds[5])};
  float4 a = boxMin[i], b = boxMax[i];
  float c[3] = {0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z)};
  uint32_t q[3];
  for (int k = 0; k < 3; ++k) {
    float ext = hi[k] - lo[k];
    float u = ext > 0.0f ? (c[k] - lo[k]) / ext : 0.0f;
    q[k] = (uint32_t)fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);
  }
  uint32_t m = (expandBits(q[0]) << 2) | (expandBits(q[1]) << 1) | expandBits(q[2]);
  keys[i] = ((unsigned long long)m << 32) | (unsigned)i;
  vals[i] = i;
}

__device__ __forceinline__ int delta(const unsigned long long* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  return __clzll((long long)(keys[i] ^ keys[j]));  // keys are unique
}

// 4. Karras 2012: internal node i covers a contiguous key range; children are internal nodes or leaves.
// child encoding here: >= 0 internal node, < 0 leaf ~sortedPosition
__global__ void lbvhHierarchy(const unsigned long long* keys, int n, int* left, int* right, int* parentInternal,
                              int* parentLeaf, uint8_t* outAxis, int base) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  int j = i + l * d;
  int dnode = delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t == 1) break;
  }
  int gamma = i + s * d + min(d, 0);
  int lo = min(i, j), hi = max(i, j);
  int lc = (lo == gamma) ? ~gamma : gamma;
  int rc = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
  left[i] = lc;
  right[i] = rc;
  if (lc >= 0) parentInternal[lc] = i; else parentLeaf[~lc] = i;
  if (rc >= 0) parentInternal[rc] = i; else parentLeaf[~rc] = i;
  if (i == 0) parentInternal[0] = -1;
  // the children differ first in key bit 63 - dnode; Morton bits (key bits 32..61) interleave x,y,z with z
  // lowest, and the left child holds the 0 side = the lower coordinate.  A tie broken by the index: unknown.
  int bit = 63 - dnode;
  outAxis[base + i] = bit >= 32 ? (uint8_t)(2 - ((bit - 32) % 3)) : (uint8_t)3;
}

// 5+6. bottom-up refit: the second thread to arrive at a node owns it; emits the node record.
// 7. each leaf also reports its depth.
__global__ void lbvhFit(const int* left, const int* right, const int* parentInternal, const int* parentLeaf,
                        const int* sortedVals, const int32_t* refs, const float4* boxMin, const float4* boxMax, int n,
                        float4* nodeMin, float4* nodeMax, int* arrived, float4* outNodes, int base, int* maxDepth) {
  int leaf = blockIdx.x * blockDim.x + threadIdx.x;
  if (leaf >= n) return;
  int depth = 1;
  int node = parentLeaf[leaf];
  while (node >= 0) {
    depth++;
    __threadfence();
    if (atomicAdd(&arrived[node], 1) == 0) {
      // first arrival: the sibling subtree is not finished; but keep counting depth to the root
      int up = parentInternal[node];
      while (up >= 0) {
        depth++;
        up = parentInternal[up];
      }
      break;
    }
    __threadfence();
    float4 mnL, mxL, mnR, mxR;
    int lc = left[node], rc = right[node];
    int lref, rref;
    if (lc >= 0) {
      mnL = nodeMin[lc];
      mxL = nodeMax[lc];
      lref = SRT_NODE_REF(lc + base);
    } else {
      int p = sortedVals[~lc];
      mnL = boxMin[p];
      mxL = boxMax[p];
      lref = refs[p];
    }
    if (rc >= 0) {
      mnR = nodeMin[rc];
      mxR = nodeMax[rc];
      rref = SRT_NODE_REF(rc + base);
    } else {
      int p = sortedVals[~rc];
      mnR = boxMin[p];
      mxR = boxMax[p];
      rref = refs[p];
    }
    float4 mn = make_float4(fminf(mnL.x, mnR.x), fminf(mnL.y, mnR.y), fminf(mnL.z, mnR.z), 0.0f);
    float4 mx = make_float4(fmaxf(mxL.x, mxR.x), fmaxf(mxL.y, mxR.y), fmaxf(mxL.z, mxR.z), 0.0f);
    nodeMin[node] = mn;
    nodeMax[node] = mx;
    outNodes[2 * (size_t)(base + node) + 0] = make_float4(mn.x, mn.y, mn.z, __int_as_float(lref));
    outNodes[2 * (size_t)(base + node) + 1] = make_float4(mx.x, mx.y, mx.z, __int_as_float(rref));
    node = parentInternal[node];
  }
  atomicMax(maxDepth, depth);
}

// one primitive: a single-object leaf like the reference's (left == right, bvh.h:67-69)
__global__ void lbvhSingle(DevScene sc, const int32_t* refs, float time0, float time1, float4* outNodes, int base) {
  float mn[3], mx[3];
  primBox(sc, refs[0], time0, time1, mn, mx);
  outNodes[2 * (size_t)base + 0] = make_float4(mn[0], mn[1], mn[2], __int_as_float(refs[0]));
  outNodes[2 * (size_t)base + 1] = make_float4(mx[0], mx[1], mx[2], __int_as_float(refs[0]));
}

}  // namespace

// Builds the tree of one world item into outNodes[base .. base + max(n-1, 1)).  dRefs: n device
// primitive refs.  Blocking.  Returns 0 and the tree depth (root = 1), or a hipError_t.
extern "C" int srt_lbvh_build(const DevScene* sc, const int32_t* dRefs, int n, float time0, float time1, float4* outNodes,
                              uint8_t* outAxis, int base, int* depthOut) {
  if (n < 1) return (int)hipErrorInvalidValue;
  if (n == 1) {
    hipLaunchKernelGGL(lbvhSingle, dim3(1), dim3(1), 0, nullptr, *sc, dRefs, time0, time1, outNodes, base);
    *depthOut = 1;
    return (int)hipDeviceSynchronize();
  }
  const size_t N = (size_t)n;
  char* pool = nullptr;
  size_t sortBytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, sortBytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                           (int*)nullptr, (int*)nullptr, N, 0, 64, nullptr);
  if (e != hipSuccess) return (int)e;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
  size_t oBoxMin = take(N * 16), oBoxMax = take(N * 16), oNodeMin = take(N * 16), oNodeMax = take(N * 16);
  size_t oKeys = take(N * 8), oKeys2 = take(N * 8), oVals = take(N * 4), oVals2 = take(N * 4);
  size_t oLeft = take(N * 4), oRight = take(N * 4), oParI = take(N * 4), oParL = take(N * 4), oArr = take(N * 4);
  size_t oBounds = take(64), oDepth = take(64), oSort = take(sortBytes);
  e = hipMalloc((void**)&pool, off);
  if (e != hipSuccess) return (int)e;
  auto P = [&](size_t o) { return pool + o; };
  int initBounds[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
  int rc = 0;
  do {
    if ((e = hipMemcpy(P(oBounds), initBounds, sizeof initBounds, hipMemcpyHostToDevice)) != hipSuccess) break;
    if ((e = hipMemset(P(oArr), 0, N * 4)) != hipSuccess) break;
    if ((e = hipMemset(P(oDepth), 0, 4)) != hipSuccess) break;
    const int B = 256, G = (n + B - 1) / B;
    hipLaunchKernelGGL(lbvhPrimBoxes, dim3(G), dim3(B), 0, nullptr, *sc, dRefs, n, time0, time1, (float4*)P(oBoxMin),
                       (float4*)P(oBoxMax), (int*)P(oBounds));
    hipLaunchKernelGGL(lbvhMorton, dim3(G), dim3(B), 0, nullptr, (const float4*)P(oBoxMin), (const float4*)P(oBoxMax), n,
                       (const int*)P(oBounds), (unsigned long long*)P(oKeys), (int*)P(oVals));
    e = rocprim::radix_sort_pairs(P(oSort), sortBytes, (unsigned long long*)P(oKeys), (unsigned long long*)P(oKeys2),
                                  (int*)P(oVals), (int*)P(oVals2), N, 0, 64, nullptr);
    if (e != hipSuccess) break;
    hipLaunchKernelGGL(lbvhHierarchy, dim3(G), dim3(B), 0, nullptr, (const unsigned long long*)P(oKeys2), n,
                       (int*)P(oLeft), (int*)P(oRight), (int*)P(oParI), (int*)P(oParL), outAxis, base);
    hipLaunchKernelGGL(lbvhFit, dim3(G), dim3(B), 0, nullptr, (const int*)P(oLeft), (const int*)P(oRight),
                       (const int*)P(oParI), (const int*)P(oParL), (const int*)P(oVals2), dRefs,
                       (const float4*)P(oBoxMin), (const float4*)P(oBoxMax), n, (float4*)P(oNodeMin), (float4*)P(oNodeMax),
                       (int*)P(oArr), outNodes, base, (int*)P(oDepth));
    if ((e = hipGetLastError()) != hipSuccess) break;
    if ((e = hipDeviceSynchronize()) != hipSuccess) break;
    if ((e = hipMemcpy(depthOut, P(oDepth), 4, hipMemcpyDeviceToHost)) != hipSuccess) break;
  } while (0);
  rc = (int)e;
  (void)hipFree(pool);
  return rc;
}

// =================================================================== PLOC (SRT_BUILDER_PLOC)
// Parallel locally-ordered clustering (Meister & Bittner 2018): the primitives, in Morton order, start as
// one cluster each; every round each cluster looks `radius` places left and right for the neighbour whose
// union with it has the smallest surface area, mutual nearest neighbours merge into a node, and the array
// is compacted in order.  It is an agglomerative build that follows the surface-area heuristic locally,
// so its trees are markedly tighter than the linear BVH's (which only looks at key prefixes), for a build
// that is still a few dozen short kernels.  Nodes are numbered from the top down in creation order
// reversed (the last merge = the root gets index 0), so children always have larger indices than parents.
namespace {

struct PlocCluster {
  float4 mn;  // xyz, w = reference (node index or primitive ref) as int bits
  float4 mx;  // xyz, w = depth of the subtree as int bits
};

__device__ __forceinline__ float unionArea(const float4& amn, const float4& amx, const float4& bmn, const float4& bmx) {
  const float dx = fmaxf(amx.x, bmx.x) - fminf(amn.x, bmn.x), dy = fmaxf(amx.y, bmx.y) - fminf(amn.y, bmn.y),
              dz = fmaxf(amx.z, bmx.z) - fminf(amn.z, bmn.z);
  return dx * dy + dy * dz + dz * dx;
}

#define PLOC_BLOCK 256
#define PLOC_MAX_RADIUS 128

// initial clusters: the primitives in Morton order
__global__ void plocInit(const int* sortedVals, const int32_t* refs, const float4* boxMin, const float4* boxMax, int n,
                         PlocCluster* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int p = sortedVals[i];
  float4 a = boxMin[p], b = boxMax[p];
  a.w = __int_as_float(refs[p]);
  b.w = __int_as_float(1);
  out[i].mn = a;
  out[i].mx = b;
}

// nearest neighbour (smallest union area) within `radius` places
__global__ __launch_bounds__(PLOC_BLOCK) void plocNearest(const PlocCluster* c, int m, int radius, int* nn) {
  __shared__ float4 sMn[PLOC_BLOCK + 2 * PLOC_MAX_RADIUS], sMx[PLOC_BLOCK + 2 * PLOC_MAX_RADIUS];
  const int first = blockIdx.x * PLOC_BLOCK - radius;
  for (int k = threadIdx.x; k < PLOC_BLOCK + 2 * radius; k += PLOC_BLOCK) {
    const int g = first + k;
    if (g >= 0 && g < m) {
      sMn[k] = c[g].mn;
      sMx[k] = c[g].mx;
    }
  }
  __syncthreads();
  const int i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
  if (i >= m) return;
  const float4 mn = sMn[threadIdx.x + radius], mx = sMx[threadIdx.x + radius];
  // Ties go to the pair partner i ^ 1 first, then to the lower position.  With "lower position" alone a run
  // of identical boxes (duplicated geometry) would merge one pair per round; this way it halves.  Some
  // mutual pair always exists: among the clusters that attain the smallest area, the lowest one either
  // pairs with its partner or with its lowest candidate b, and b answers with it unless b pairs with b ^ 1.
  float best = 3.0e38f;
  int bestJ = -1;
  const int lo = max(0, i - radius), hi = min(m - 1, i + radius);
  for (int j = lo; j <= hi; ++j) {
    if (j == i) continue;
    const float a = unionArea(mn, mx, sMn[j - first], sMx[j - first]);
    if (a < best || (a == best && j == (i ^ 1))) {
      best = a;
      bestJ = j;
    }
  }
  nn[i] = bestJ;  // -1 only when m == 1
}

// flags for the in-order compaction: high word = the cluster survives (it is not the right-hand partner of
// a merge), low word = it is the left-hand partner (one new node)
__global__ void plocDecide(const int* nn, int m, unsigned long long* flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int j = nn[i];
  const bool mutual = j >= 0 && nn[j] == i;
  const bool leader = mutual && i < j, absorbed = mutual && i > j;
  flags[i] = ((unsigned long long)(absorbed ? 0 : 1) << 32) | (unsigned long long)(leader ? 1 : 0);
}

// emit this round's nodes and the next round's cluster array
__global__ void plocMerge(const PlocCluster* c, const int* nn, const unsigned long long* flags, const unsigned long long* scanned,
                          int m, int n, int nodesBefore, int base, float4* outNodes, uint8_t* outAxis, PlocCluster* next) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const unsigned long long f = flags[i], sc = scanned[i];
  if (!(f >> 32)) return;  // absorbed by its partner
  const int pos = (int)(sc >> 32);
  PlocCluster me = c[i];
  if (f & 1ull) {
    const PlocCluster other = c[nn[i]];
    // node index: creation order reversed, so that the last node made (the root) is 0
    const int node = (n - 2) - (nodesBefore + (int)(sc & 0xffffffffull));
    // left child = lower centroid along the axis where the two centroids differ most (near-child-first
    // traversal reads the axis; 3 = no usable axis)
    const float cx = (other.mn.x + other.mx.x) - (me.mn.x + me.mx.x), cy = (other.mn.y + other.mx.y) - (me.mn.y + me.mx.y),
                cz = (other.mn.z + other.mx.z) - (me.mn.z + me.mx.z);
    int axis = fabsf(cx) >= fabsf(cy) ? (fabsf(cx) >= fabsf(cz) ? 0 : 2) : (fabsf(cy) >= fabsf(cz) ? 1 : 2);
    const float d = axis == 0 ? cx : (axis == 1 ? cy : cz);
    const bool meFirst = d >= 0.0f;
    if (d == 0.0f) axis = 3;
    const float4 mn = make_float4(fminf(me.mn.x, other.mn.x), fminf(me.mn.y, other.mn.y), fminf(me.mn.z, other.mn.z), 0.0f);
    const float4 mx = make_float4(fmaxf(me.mx.x, other.mx.x), fmaxf(me.mx.y, other.mx.y), fmaxf(me.mx.z, other.mx.z), 0.0f);
    const float4 l = meFirst ? me.mn : other.mn, r = meFirst ? other.mn : me.mn;
    outNodes[2 * (size_t)(base + node) + 0] = make_float4(mn.x, mn.y, mn.z, l.w);
    outNodes[2 * (size_t)(base + node) + 1] = make_float4(mx.x, mx.y, mx.z, r.w);
    outAxis[base + node] = (uint8_t)axis;
    const int depth = max(__float_as_int(me.mx.w), __float_as_int(other.mx.w)) + 1;
    me.mn = make_float4(mn.x, mn.y, mn.z, __int_as_float(SRT_NODE_REF(base + node)));
    me.mx = make_float4(mx.x, mx.y, mx.z, __int_as_float(depth));
  }
  next[pos] = me;
}

}  // namespace

// The closest-hit traversal's node records: both children's boxes in one 64-byte record (DevScene::nodes2).
// A node child's box is that node's own box; a primitive child's box is computed by the reference's
// boundingBox rules over [time0, time1] (the widest range of the scene's items: a larger box is still correct,
// the traversal only needs boxes that contain their primitives).
namespace {
__global__ void pairNodes(DevScene sc, float time0, float time1, float4* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sc.numNodes) return;
  const float4 n0 = sc.nodes[2 * (size_t)i], n1 = sc.nodes[2 * (size_t)i + 1];
  const int refs[2] = {__float_as_int(n0.w), __float_as_int(n1.w)};
  float4 rec[4];
  for (int c = 0; c < 2; ++c) {
    float mn[3], mx[3];
    int ref = refs[c];
    if (ref >= 0) {
      const int j = SRT_NODE_INDEX(ref);
      const float4 c0 = sc.nodes[2 * (size_t)j], c1 = sc.nodes[2 * (size_t)j + 1];
      mn[0] = c0.x; mn[1] = c0.y; mn[2] = c0.z;
      mx[0] = c1.x; mx[1] = c1.y; mx[2] = c1.z;
      ref = j << 6;  // byte offset of the child's 64-byte record
    } else if (ref == SRT_REF_DONE || ref == 0) {
      mn[0] = mn[1] = mn[2] = 1.0f;  // unused slot (padding record): an empty box
      mx[0] = mx[1] = mx[2] = -1.0f;
    } else {
      primBox(sc, ref, time0, time1, mn, mx);
    }
    rec[2 * c + 0] = make_float4(mn[0], mn[1], mn[2], c == 0 ? __int_as_float(ref) : 0.0f);
    rec[2 * c + 1] = make_float4(mx[0], mx[1], mx[2], 0.0f);
    if (c == 1) rec[1].w = __int_as_float(ref);
  }
  for (int k = 0; k < 4; ++k) out[4 * (size_t)i + k] = rec[k];
}
}  // namespace

// ---- wide records of the closest-hit traversal: FOUR boxes per 128-byte record.
// nodes4[i] = the grandchildren of binary node i (a child that is a primitive stands for itself): slots k = 0..3 hold
// (min_k.xyz, reference_k) at +16k and (max_k.xyz, -) at +64 + 16k; node references are byte offsets into this array
// (index * 128), primitive references as everywhere, SRT_REF_DONE marks an unused slot.  Built from the paired records
// (pairNodes above) for every node; the traversal only ever reaches the root and the nodes that are somebody's slot.
// One visit = one full 128-byte line and four box tests; a ray crosses half as many records as with the 64-byte ones.
namespace {
__global__ void wideNodes(const float4* nodes2, int numNodes, float4* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= numNodes) return;
  float4 mn[4], mx[4];
  int n = 0;
  auto put = [&](float4 lo, float4 hi, int ref) {
    if (ref == SRT_REF_DONE || ref == 0) return;  // padding
    for (int k = 0; k < n; ++k)
      if (__float_as_int(mn[k].w) == ref) return;  // a single-object leaf names its object twice (bvh.h:67-69)
    mn[n] = make_float4(lo.x, lo.y, lo.z, __int_as_float(ref));
    mx[n] = make_float4(hi.x, hi.y, hi.z, 0.0f);
    n++;
  };
  auto wideRef = [](int ref) { return ref >= 0 ? (ref >> 6) << 7 : ref; };  // node: byte offset in nodes2 -> in nodes4
  const float4* me = nodes2 + 4 * (size_t)i;
  const int childRef[2] = {__float_as_int(me[0].w), __float_as_int(me[1].w)};
  for (int c = 0; c < 2; ++c) {
    const int ref = childRef[c];
    if (ref >= 0 && !(c == 1 && ref == childRef[0])) {  // a node: its two children take slots
      const float4* ch = nodes2 + 4 * (size_t)(ref >> 6);
      put(ch[0], ch[1], wideRef(__float_as_int(ch[0].w)));
      put(ch[2], ch[3], wideRef(__float_as_int(ch[1].w)));
    } else if (ref < 0) {  // a primitive: its own slot
      put(me[2 * c], me[2 * c + 1], ref);
    }
  }
  for (int k = n; k < 4; ++k) {
    mn[k] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(SRT_REF_DONE));
    mx[k] = make_float4(-1.0f, -1.0f, -1.0f, 0.0f);
  }
  for (int k = 0; k < 4; ++k) {
    out[8 * (size_t)i + k] = mn[k];
    out[8 * (size_t)i + 4 + k] = mx[k];
  }
}
}  // namespace

extern "C" int srt_wide_nodes(const float4* nodes2, int numNodes, float4* out) {
  if (numNodes <= 0) return 0;
  hipLaunchKernelGGL(wideNodes, dim3((numNodes + 255) / 256), dim3(256), 0, nullptr, nodes2, numNodes, out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  return (int)e;
}

extern "C" int srt_pair_nodes(const DevScene* sc, float time0, float time1, float4* out) {
  if (sc->numNodes <= 0) return 0;
  hipLaunchKernelGGL(pairNodes, dim3((sc->numNodes + 255) / 256), dim3(256), 0, nullptr, *sc, time0, time1, out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  return (int)e;
}

// Same contract as srt_lbvh_build.  radius: clusters examined on either side (1..128).
extern "C" int srt_ploc_build(const DevScene* sc, const int32_t* dRefs, int n, float time0, float time1, float4* outNodes,
                              uint8_t* outAxis, int base, int radius, int* depthOut) {
  if (n < 1) return (int)hipErrorInvalidValue;
  if (n == 1) {
    hipLaunchKernelGGL(lbvhSingle, dim3(1), dim3(1), 0, nullptr, *sc, dRefs, time0, time1, outNodes, base);
    *depthOut = 1;
    return (int)hipDeviceSynchronize();
  }
  radius = radius < 1 ? 1 : (radius > PLOC_MAX_RADIUS ? PLOC_MAX_RADIUS : radius);
  const size_t N = (size_t)n;
  char* pool = nullptr;
  size_t sortBytes = 0, scanBytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, sortBytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                           (int*)nullptr, (int*)nullptr, N, 0, 64, nullptr);
  if (e != hipSuccess) return (int)e;
  e = rocprim::exclusive_scan(nullptr, scanBytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, 0ull, N,
                              rocprim::plus<unsigned long long>(), nullptr);
  if (e != hipSuccess) return (int)e;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
  const size_t oBoxMin = take(N * 16), oBoxMax = take(N * 16), oKeys = take(N * 8), oKeys2 = take(N * 8), oVals = take(N * 4),
               oVals2 = take(N * 4), oBounds = take(64), oClA = take(N * sizeof(PlocCluster)), oClB = take(N * sizeof(PlocCluster)),
               oNn = take(N * 4), oFlags = take(N * 8), oScan = take(N * 8), oTemp = take(std::max(sortBytes, scanBytes));
  e = hipMalloc((void**)&pool, off);
  if (e != hipSuccess) return (int)e;
  auto P = [&](size_t o) { return pool + o; };
  const int initBounds[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
  do {
    if ((e = hipMemcpy(P(oBounds), initBounds, sizeof initBounds, hipMemcpyHostToDevice)) != hipSuccess) break;
    const int B = PLOC_BLOCK, G = (n + B - 1) / B;
    hipLaunchKernelGGL(lbvhPrimBoxes, dim3(G), dim3(B), 0, nullptr, *sc, dRefs, n, time0, time1, (float4*)P(oBoxMin),
                       (float4*)P(oBoxMax), (int*)P(oBounds));
    hipLaunchKernelGGL(lbvhMorton, dim3(G), dim3(B), 0, nullptr, (const float4*)P(oBoxMin), (const float4*)P(oBoxMax), n,
                       (const int*)P(oBounds), (unsigned long long*)P(oKeys), (int*)P(oVals));
    e = rocprim::radix_sort_pairs(P(oTemp), sortBytes, (unsigned long long*)P(oKeys), (unsigned long long*)P(oKeys2),
                                  (int*)P(oVals), (int*)P(oVals2), N, 0, 64, nullptr);
    if (e != hipSuccess) break;
    PlocCluster *cur = (PlocCluster*)P(oClA), *nxt = (PlocCluster*)P(oClB);
    hipLaunchKernelGGL(plocInit, dim3(G), dim3(B), 0, nullptr, (const int*)P(oVals2), dRefs, (const float4*)P(oBoxMin),
                       (const float4*)P(oBoxMax), n, cur);
    int m = n, nodes = 0;
    // every round merges at least the globally closest mutual pair, so m strictly decreases
    for (int round = 0; m > 1 && round < 4 * 64 + n; ++round) {
      const int g = (m + B - 1) / B;
      hipLaunchKernelGGL(plocNearest, dim3(g), dim3(B), 0, nullptr, cur, m, radius, (int*)P(oNn));
      hipLaunchKernelGGL(plocDecide, dim3(g), dim3(B), 0, nullptr, (const int*)P(oNn), m, (unsigned long long*)P(oFlags));
      size_t sb = scanBytes;
      e = rocprim::exclusive_scan(P(oTemp), sb, (unsigned long long*)P(oFlags), (unsigned long long*)P(oScan), 0ull, (size_t)m,
                                  rocprim::plus<unsigned long long>(), nullptr);
      if (e != hipSuccess) break;
      hipLaunchKernelGGL(plocMerge, dim3(g), dim3(B), 0, nullptr, cur, (const int*)P(oNn), (const unsigned long long*)P(oFlags),
                         (const unsigned long long*)P(oScan), m, n, nodes, base, outNodes, outAxis, nxt);
      unsigned long long lastScan = 0, lastFlag = 0;
      if ((e = hipMemcpy(&lastScan, P(oScan) + (size_t)(m - 1) * 8, 8, hipMemcpyDeviceToHost)) != hipSuccess) break;
      if ((e = hipMemcpy(&lastFlag, P(oFlags) + (size_t)(m - 1) * 8, 8, hipMemcpyDeviceToHost)) != hipSuccess) break;
      const unsigned long long total = lastScan + lastFlag;
      const int merged = (int)(total & 0xffffffffull), kept = (int)(total >> 32);
      if (merged < 1 || kept != m - merged) {
        e = hipErrorUnknown;  // cannot happen: the closest pair is always mutual
        break;
      }
      nodes += merged;
      m = kept;
      std::swap(cur, nxt);
    }
    if (e != hipSuccess) break;
    if ((e = hipGetLastError()) != hipSuccess) break;
    if (m != 1 || nodes != n - 1) {
      e = hipErrorUnknown;
      break;
    }
    PlocCluster root;
    if ((e = hipMemcpy(&root, cur, sizeof root, hipMemcpyDeviceToHost)) != hipSuccess) break;
    memcpy(depthOut, &root.mx.w, 4);
  } while (0);
  (void)hipFree(pool);
  return (int)e;
}
