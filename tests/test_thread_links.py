"""Host logic of the stackless walks (csrc/srt_thread.h, no GPU): the thread links srtUploadScene builds -- the 16-bit
form the LDS-resident-tree kernels walk and the path-pool kernel's hybrid records (32-bit references, resident nodes
first) -- must make a stackless walk visit exactly what bvhNode::hit's recursion visits (bvh.h:97-105: this node's
box; on a hit the left child, then the right child, always), for every pattern of box hits and misses.

The walks below restate the kernels' node and primitive steps (srt_wavefront.hip `visit` / `popNext`, srt_kernels.hip
LDSTREE) on the host; the box test is replaced by a seeded coin per node, which covers every hit / miss combination a
ray could produce and many it could not."""
import numpy as np
import pytest

DONE16 = -32768          # 0x8000 sign-extended
DONE_W = -(1 << 29)


def _forest(rng, leaves_per_tree, world_prims=0):
    """Random world list in the flattened device format: pre-order node arrays of random binary trees whose leaves
    hold one object (left == right, as bvh.h:64-66 builds them) or two (neighbouring triangles, far-apart triangles, a
    triangle and a sphere, two spheres), plus bare primitives in the world list."""
    nodes = []  # [lo(3), left, hi(3), right]
    counters = {"tri": 0, "sph": 0}

    def new_prim(kind):
        i = counters[kind]
        counters[kind] += 1
        return ~((i << 1) | (1 if kind == "sph" else 0))

    def build(n_leaves, size):
        me = len(nodes)
        nodes.append(None)
        lo = (float(me), 0.0, 0.0)
        hi = (float(me) + size, size, size)
        if n_leaves == 1:
            shape = rng.integers(0, 5)
            if shape == 0:
                l = r = new_prim("tri" if rng.integers(0, 4) else "sph")
            elif shape == 1:
                l, r = new_prim("tri"), new_prim("tri")          # neighbours in the triangle array
            elif shape == 2:
                l = new_prim("tri"); counters["tri"] += int(rng.integers(1, 4)); r = new_prim("tri")  # not neighbours
            elif shape == 3:
                l, r = (new_prim("tri"), new_prim("sph")) if rng.integers(0, 2) else (new_prim("sph"), new_prim("tri"))
            else:
                l, r = new_prim("sph"), new_prim("sph")
            nodes[me] = (lo, l, hi, r)
            return me
        k = int(rng.integers(1, n_leaves))
        left = build(k, size * float(rng.uniform(0.3, 0.9)))
        right = build(n_leaves - k, size * float(rng.uniform(0.3, 0.9)))
        nodes[me] = (lo, left * 32, hi, right * 32)
        return me

    world = []
    for n_leaves in leaves_per_tree:
        world.append(build(n_leaves, 1000.0 * float(rng.uniform(0.5, 1.0))) * 32)
        for _ in range(world_prims):
            world.append(new_prim("tri" if rng.integers(0, 2) else "sph"))
    arr = np.zeros((len(nodes), 8), np.float32)
    refs = arr.view(np.int32)
    for i, (lo, l, hi, r) in enumerate(nodes):
        arr[i, 0:3], arr[i, 4:7] = lo, hi
        refs[i, 3], refs[i, 7] = l, r
    return arr, np.asarray(world, np.int32), counters["tri"], counters["sph"]


def _coin(seed, node_id):
    x = (node_id * 2654435761 + seed * 40503) & 0xffffffff
    x ^= x >> 15
    x = (x * 2246822519) & 0xffffffff
    return ((x >> 13) & 3) != 0  # a box is hit three times in four


def _recursion(arr, world, seed):
    """bvh.h:97-105 and hittablelist.h:33-47 (every object of the list, in order)."""
    refs = arr.view(np.int32)
    out = []
    for w in world:
        todo = [int(w)]
        while todo:
            r = todo.pop()
            if r < 0:
                out.append(("prim", r))
                continue
            i = r >> 5
            out.append(("node", i))
            if not _coin(seed, i):
                continue
            l, rr = int(refs[i, 3]), int(refs[i, 7])
            if l >= 0:
                todo.append(rr)
                todo.append(l)
            else:
                out.append(("prim", l))
                if rr != l:
                    out.append(("prim", rr))
    return out


def _walk16(arr, links, world, seed):
    """srt_kernels.hip LDSTREE / srt_wavefront.hip (whole tree in LDS): cur, link pair, popNext."""
    refs = arr.view(np.int32)
    out = []
    for w in world:
        cur = int(w) >> 5 if w >= 0 else int(w)
        link = (DONE16 << 16) | (DONE16 & 0xffff)
        while cur != DONE16:
            if cur >= 0:
                out.append(("node", cur))
                link = int(links[cur])
                taken = int(refs[cur, 3])
                taken = taken >> 5 if taken >= 0 else taken
                cur = taken if _coin(seed, cur) else link >> 16
            else:
                out.append(("prim", cur))
                low = link & 0xffff
                cur = low - 0x10000 if low & 0x8000 else low
                link >>= 16
            assert len(out) < 10 * len(arr) + 100, "the walk does not end"
    return out


def _walk_hybrid(wf, world_wf, second, seed, ids):
    """srt_wavefront.hip HYBRID: visit / popNext over the renumbered 32-bit records; `ids` maps a record to the original
    node (its box's lo.x)."""
    refs = wf.view(np.int32)
    out = []
    for w in world_wf:
        cur, link = int(w), -(1 << 31)
        while cur != DONE_W:
            if cur >= 0:
                out.append(("node", ids[cur]))
                link = int(refs[cur, 7])
                cur = int(refs[cur, 3]) if _coin(seed, ids[cur]) else link >> 2
            else:
                out.append(("prim", cur))
                follows, nxt = link & 3, link >> 2
                if follows == 1:
                    nxt = cur - 2
                elif follows == 2:
                    nxt = int(second[~cur])
                link &= ~3
                cur = nxt
            assert len(out) < 10 * len(wf) + 100, "the walk does not end"
    return out


def _links16(dev, arr, world, ntri, nsph):
    out = np.zeros(len(arr), np.int32)
    rc = dev.lib.srtTestThreadLinks16(arr.ctypes.data, len(arr), world.ctypes.data, len(world), ntri, nsph, out.ctypes.data)
    assert rc in (0, 1)
    return out if rc else None


def _hybrid(dev, arr, world, ntri, nsph, cap):
    wf = np.zeros_like(arr)
    ww = np.zeros(len(world), np.int32)
    second = np.zeros(2 * max(ntri, nsph) + 2, np.int32)
    rc = dev.lib.srtTestHybridRecords(arr.ctypes.data, len(arr), world.ctypes.data, len(world), ntri, nsph, cap, wf.ctypes.data, ww.ctypes.data,
                                      second.ctypes.data)
    assert rc >= 0
    return (rc, wf, ww, second) if rc else None


@pytest.fixture(scope="module")
def hooks(dev):
    return dev  # srtTestThreadLinks16 / srtTestHybridRecords: argument types in hipdev.py


@pytest.mark.parametrize("case", range(12))
def test_stackless_walks_visit_what_the_recursion_visits(hooks, case):
    rng = np.random.default_rng(100 + case)
    trees = [int(rng.integers(1, 60)) for _ in range(1 + case % 3)]
    arr, world, ntri, nsph = _forest(rng, trees, world_prims=case % 2)
    links = _links16(hooks, arr, world, ntri, nsph)
    assert links is not None
    n = len(arr)
    for cap in (1, 2, max(1, n // 3), n - 1 if n > 1 else 1, n + 5):
        h = _hybrid(hooks, arr, world, ntri, nsph, cap)
        assert h is not None
        resident, wf, ww, second = h
        assert resident == min(cap, n)
        ids = [int(x) for x in wf[:, 0]]  # lo.x of a record = the original index of its node
        assert sorted(ids) == list(range(n))
        # boxes travel with their node; the resident set is closed upward: a resident node's parent is resident
        new_of = {orig: k for k, orig in enumerate(ids)}
        refs = arr.view(np.int32)
        for orig in range(n):
            k = new_of[orig]
            assert np.array_equal(wf[k, 0:3], arr[orig, 0:3]) and np.array_equal(wf[k, 4:7], arr[orig, 4:7])
            if refs[orig, 3] >= 0:
                for child in (int(refs[orig, 3]) >> 5, int(refs[orig, 7]) >> 5):
                    assert not (new_of[child] < resident and k >= resident), (orig, child)
                # a node's first child is the next record when both are on the same side of the boundary
                c = new_of[int(refs[orig, 3]) >> 5]
                if (c < resident) == (k < resident):
                    assert c == k + 1
        for seed in range(6):
            want = _recursion(arr, world, seed)
            assert _walk16(arr, links, world, seed) == want, (case, seed)
            assert _walk_hybrid(wf, ww, second, seed, ids) == want, (case, cap, seed)


def test_no_threaded_form_for_other_trees(hooks):
    """A node with one node child and one primitive child (a caller-built tree may have them, bvh.h builds none), a node
    reached twice, references beyond 15 bits: no links -- the kernels then walk with a stack."""
    rng = np.random.default_rng(7)
    arr, world, ntri, nsph = _forest(rng, [8])
    refs = arr.view(np.int32)
    mixed = arr.copy()
    inner = next(i for i in range(len(arr)) if refs[i, 3] >= 0)
    mixed.view(np.int32)[inner, 7] = ~0
    assert _links16(hooks, mixed, world, ntri, nsph) is None and _hybrid(hooks, mixed, world, ntri, nsph, 4) is None
    twice = arr.copy()
    twice.view(np.int32)[inner, 7] = refs[inner, 3]
    assert _links16(hooks, twice, world, ntri, nsph) is None and _hybrid(hooks, twice, world, ntri, nsph, 4) is None
    assert _links16(hooks, arr, world, 20000, nsph) is None      # 2 x 20000 does not fit 15 bits
    assert _hybrid(hooks, arr, world, 20000, nsph, 4) is not None  # 32-bit references do not mind


def test_real_tree_walks(hooks, srt):
    """The headline mesh's tree as the host builder makes it (bvh.h:55-95 with the reference's generator): both threaded
    forms against the recursion, triangles numbered by first appearance as srtUploadScene numbers them."""
    nodes, _ = hooks.build_bvh_host(srt.scenes.scene_masterchief())
    n = len(nodes)
    arr = np.zeros((n, 8), np.float32)
    refs = arr.view(np.int32)
    order = {}
    for i in range(n):
        arr[i, 0:3], arr[i, 4:7] = nodes[i]["bmin"], nodes[i]["bmax"]
        arr[i, 0] = i  # the walks identify a record by lo.x
        for col, c in ((3, int(nodes[i]["left"])), (7, int(nodes[i]["right"]))):
            refs[i, col] = c * 32 if c >= 0 else ~(order.setdefault(~c, len(order)) << 1)
    world = np.asarray([0], np.int32)
    links = _links16(hooks, arr, world, len(order), 0)
    h = _hybrid(hooks, arr, world, len(order), 0, 1500)
    assert links is not None and h is not None and h[0] == 1500
    _, wf, ww, second = h
    ids = [int(x) for x in wf[:, 0]]
    for seed in [s for s in range(40) if _coin(s, 0)][:3]:  # coins that let the walk past the root
        want = _recursion(arr, world, seed)
        assert len(want) > 100
        assert _walk16(arr, links, world, seed) == want
        assert _walk_hybrid(wf, ww, second, seed, ids) == want
