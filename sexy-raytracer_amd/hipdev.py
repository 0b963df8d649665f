"""ctypes binding of csrc/libsrt_hip.so, the C-ABI library declared in include/srt_hip.h.

Importing this module loads the HIP extension and raises if it is not built: there is
no CPU fallback for the hot path."""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsrt_hip.so")
if os.environ.get("SRT_HIP_LIB"):  # measurement tools only: an experimental build of the same library
    LIB_PATH = os.path.abspath(os.environ["SRT_HIP_LIB"])

if not os.path.exists(LIB_PATH):
    raise ImportError("HIP extension not built: %s is missing (run __graft_entry__.build() or "
                      "`make -C sexy-raytracer_amd/csrc`)" % LIB_PATH)

# PyTorch-ROCm wheels carry their own libamdhip64 / libhsa-runtime64 (same sonames as /opt/rocm's, which this library is
# linked against).  One process gets ONE of them -- whichever is loaded first -- and torch does not find the GPU through
# /opt/rocm's copy ("No HIP GPUs are available").  The tests and bench.py take their device buffers from torch, so torch
# goes first when it is there; the library runs on either copy.  (C / C++ callers never meet this.)
try:
    import torch  # noqa: F401
except ImportError:
    pass

lib = C.CDLL(LIB_PATH)

EXPORTS = ["srtCreate", "srtDestroy", "srtLastError", "srtMakeCamera", "srtHostRandomFloat", "srtHostRandomReset",
           "srtUploadScene", "srtSetCamera", "srtBuildBvh", "srtGetBvh", "srtGetBvhDepth", "srtNumTiles", "srtNumLocalTiles", "srtDefaultSppChunks", "srtPlanSppChunks",
           "srtRenderTiles", "srtResolveTiles", "srtRenderImage", "srtTraceRays", "srtScatterRays",
           "srtCommGetUniqueId", "srtCommInit", "srtGatherTiles", "srtRenderImageRanks", "srtCommDestroy",
           "srtLastKernelMs", "srtGetStats", "srtDeviceInfo"]
# include/srt_hip_test.h: test hooks and diagnostics, not part of the drop-in boundary
TEST_EXPORTS = ["srtScatterTest", "srtSetTunable", "srtGetTunable", "srtGetShadeProfile", "srtGetWfProfile", "srtGetLaunchInfo", "srtRenderAov",
                "srtTestThreadLinks16", "srtTestHybridRecords"]

_vp = C.c_void_p
lib.srtCreate.argtypes = [C.c_int, C.POINTER(_vp)]
lib.srtDestroy.argtypes = [_vp]
lib.srtLastError.argtypes = [_vp]
lib.srtLastError.restype = C.c_char_p
lib.srtMakeCamera.argtypes = [C.POINTER(abi.SrtCameraParams), C.POINTER(abi.SrtCamera)]
lib.srtHostRandomFloat.restype = C.c_float
lib.srtHostRandomReset.restype = None
lib.srtUploadScene.argtypes = [_vp, C.POINTER(abi.SrtSceneDesc)]
lib.srtSetCamera.argtypes = [_vp, C.POINTER(abi.SrtCamera)]
lib.srtBuildBvh.argtypes = [C.POINTER(abi.SrtSceneDesc), C.c_int32, _vp, C.c_int32, C.POINTER(C.c_int32),
                            C.POINTER(C.c_int32)]
lib.srtGetBvh.argtypes = [_vp, C.c_int32, _vp, C.c_int32, C.POINTER(C.c_int32)]
lib.srtGetBvhDepth.argtypes = [_vp, C.POINTER(C.c_int32)]
lib.srtNumTiles.argtypes = [C.c_int32, C.c_int32]
lib.srtNumTiles.restype = C.c_int32
lib.srtNumLocalTiles.argtypes = [C.c_int32, C.c_int32, C.c_int32]
lib.srtNumLocalTiles.restype = C.c_int32
lib.srtDefaultSppChunks.argtypes = [C.c_int32]
lib.srtDefaultSppChunks.restype = C.c_int32
lib.srtPlanSppChunks.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32]
lib.srtPlanSppChunks.restype = C.c_int32
lib.srtRenderTiles.argtypes = [_vp, C.POINTER(abi.SrtRenderParams), _vp, _vp]
lib.srtResolveTiles.argtypes = [_vp, C.POINTER(abi.SrtRenderParams), _vp, _vp, _vp, _vp]
lib.srtRenderImage.argtypes = [_vp, C.POINTER(abi.SrtRenderParams), _vp, _vp]
lib.srtTraceRays.argtypes = [_vp, _vp, C.c_int64, _vp, C.c_int32]
lib.srtCommGetUniqueId.argtypes = [_vp]
lib.srtCommInit.argtypes = [_vp, _vp, C.c_int32, C.c_int32]
lib.srtGatherTiles.argtypes = [_vp, C.POINTER(abi.SrtRenderParams), _vp, _vp, _vp]
lib.srtRenderImageRanks.argtypes = [_vp, C.POINTER(abi.SrtRenderParams), _vp, _vp]
lib.srtCommDestroy.argtypes = [_vp]
lib.srtScatterTest.argtypes = [_vp, _vp, _vp, C.c_int32, C.c_uint64, _vp]
lib.srtScatterRays.argtypes = [_vp, _vp, _vp, C.c_int32, C.c_uint64, _vp]
lib.srtSetTunable.argtypes = [_vp, C.c_char_p, C.c_int32]
lib.srtGetTunable.argtypes = [_vp, C.c_char_p, C.POINTER(C.c_int32)]
lib.srtGetShadeProfile.argtypes = [_vp, _vp]
lib.srtGetWfProfile.argtypes = [_vp, _vp]
lib.srtGetLaunchInfo.argtypes = [_vp, _vp]
lib.srtTestThreadLinks16.argtypes = [_vp, C.c_int32, _vp, C.c_int32, C.c_int32, C.c_int32, _vp]
lib.srtTestHybridRecords.argtypes = [_vp, C.c_int32, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp]
lib.srtTestThreadLinks16.restype = lib.srtTestHybridRecords.restype = C.c_int32
lib.srtRenderAov.argtypes = [_vp, C.POINTER(abi.SrtRenderParams), C.c_int32, _vp]
lib.srtLastKernelMs.argtypes = [_vp, C.POINTER(C.c_float)]
lib.srtGetStats.argtypes = [_vp, C.POINTER(abi.SrtStats)]
lib.srtDeviceInfo.argtypes = [_vp, C.c_char_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]


class SrtError(RuntimeError):
    pass


def make_camera(params):
    """camera ctor arithmetic (camera.h:10-38) on the host; needs no GPU."""
    cam = abi.SrtCamera()
    if lib.srtMakeCamera(C.byref(params), C.byref(cam)):
        raise SrtError("srtMakeCamera failed")
    return cam


def host_random_reset():
    lib.srtHostRandomReset()


def host_random_float():
    return float(lib.srtHostRandomFloat())


def num_tiles(w, h):
    return int(lib.srtNumTiles(w, h))


def num_local_tiles(w, h, stride):
    return int(lib.srtNumLocalTiles(w, h, stride))


def default_spp_chunks(spp):
    return int(lib.srtDefaultSppChunks(spp))


def plan_spp_chunks(width, height, spp, spp_chunks=0):
    """The chunk count a render of this size uses (-1: an explicit count that does not fit); include/srt_hip.h."""
    return int(lib.srtPlanSppChunks(width, height, spp, spp_chunks))


def _reset_generator_for(scene_builder):
    """The process-global generator as the reference's process would have it when the world's bvhNode is
    constructed: fresh (globals.h:31-32), minus the draws the scene's own construction code took
    (scenes.scene_sphere_field records them)."""
    host_random_reset()
    for _ in range(int(getattr(scene_builder, "global_rng_draws", 0))):
        lib.srtHostRandomFloat()


def build_bvh_host(scene_builder, item=0, reset_rng=True):
    """bvh.h:55-95 on the host, no GPU: returns (nodes, traversal stack depth)."""
    desc = scene_builder.desc()
    n, sd = C.c_int32(0), C.c_int32(0)
    if reset_rng:
        _reset_generator_for(scene_builder)
    if lib.srtBuildBvh(C.byref(desc), item, None, 0, C.byref(n), C.byref(sd)):
        raise SrtError("srtBuildBvh failed")
    if reset_rng:  # the sizing call consumed generator draws: rebuild from the same state
        _reset_generator_for(scene_builder)
    nodes = np.zeros(n.value, abi.NODE_DTYPE)
    if lib.srtBuildBvh(C.byref(desc), item, nodes.ctypes.data, n.value, C.byref(n), C.byref(sd)):
        raise SrtError("srtBuildBvh failed")
    return nodes, sd.value


def comm_unique_id():
    """128 opaque bytes from RCCL (rank 0 calls this and hands them to the other ranks)."""
    buf = C.create_string_buffer(128)
    if lib.srtCommGetUniqueId(buf):
        raise SrtError("srtCommGetUniqueId failed")
    return buf.raw


class Context:
    """One per GPU (glDevice's role, gl.h:16-40)."""

    def __init__(self, device=0):
        h = _vp()
        if lib.srtCreate(device, C.byref(h)):
            raise SrtError("srtCreate(%d) failed: no usable HIP device" % device)
        self.h = h
        self._scene_keep = None

    def close(self):
        # `lib` is already gone when a context is collected at interpreter shutdown
        if getattr(self, "h", None) and lib is not None:
            lib.srtDestroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc):
        if rc:
            raise SrtError(lib.srtLastError(self.h).decode())

    def upload_scene(self, scene_builder, reset_rng=True):
        """reset_rng: start from the process-global generator of a new process of the reference
        (globals.h:31-32), advanced past the draws the scene's construction took."""
        if reset_rng:
            _reset_generator_for(scene_builder)
        desc = scene_builder.desc()
        self._scene_keep = (scene_builder, desc)
        self._check(lib.srtUploadScene(self.h, C.byref(desc)))

    def set_camera(self, cam):
        self._check(lib.srtSetCamera(self.h, C.byref(cam)))
        self._camera_bytes = bytes(cam)

    def fingerprint(self):
        """sha1 over the uploaded scene description and the camera: what a checkpoint of accumulated samples
        belongs to (progressive.py)."""
        import hashlib
        h = hashlib.sha1()
        if self._scene_keep is not None:
            for a in self._scene_keep[0]._keep:
                h.update(bytes(memoryview(a)) if not hasattr(a, "tobytes") else a.tobytes())
        h.update(getattr(self, "_camera_bytes", b""))
        return h.hexdigest()

    def bvh(self, item=0):
        n = C.c_int32(0)
        self._check(lib.srtGetBvh(self.h, item, None, 0, C.byref(n)))
        nodes = np.zeros(n.value, abi.NODE_DTYPE)
        self._check(lib.srtGetBvh(self.h, item, nodes.ctypes.data, n.value, C.byref(n)))
        return nodes

    def bvh_depth(self):
        d = C.c_int32(0)
        self._check(lib.srtGetBvhDepth(self.h, C.byref(d)))
        return d.value

    def render_image(self, params, want_accum=True, want_rgba=True):
        W, H = params.imageWidth, params.imageHeight
        accum = np.zeros((H, W, 4), np.float32) if want_accum else None
        rgba = np.zeros((H, W, 4), np.uint8) if want_rgba else None
        self._check(lib.srtRenderImage(self.h, C.byref(params), accum.ctypes.data if want_accum else None,
                                       rgba.ctypes.data if want_rgba else None))
        return accum, rgba

    def render_tiles(self, params, d_accum_ptr, stream=None):
        self._check(lib.srtRenderTiles(self.h, C.byref(params), d_accum_ptr, stream))

    def resolve_tiles(self, params, d_gathered_ptr, d_rgba_ptr=None, d_accum_image_ptr=None, stream=None):
        self._check(lib.srtResolveTiles(self.h, C.byref(params), d_gathered_ptr, d_rgba_ptr, d_accum_image_ptr, stream))

    def comm_init(self, unique_id, num_ranks, rank):
        self._check(lib.srtCommInit(self.h, C.c_char_p(unique_id), num_ranks, rank))

    def gather_tiles(self, params, d_local_ptr, d_gathered_ptr=None, stream=None):
        """The path's one collective: ncclGather of the ranks' tile buffers to rank 0 (srt_comm.cpp)."""
        self._check(lib.srtGatherTiles(self.h, C.byref(params), d_local_ptr, d_gathered_ptr, stream))

    def render_image_ranks(self, params, want_accum=True, want_rgba=True):
        """Collective blocking render across the communicator's ranks; rank 0 gets the image."""
        W, H = params.imageWidth, params.imageHeight
        accum = np.zeros((H, W, 4), np.float32) if want_accum else None
        rgba = np.zeros((H, W, 4), np.uint8) if want_rgba else None
        self._check(lib.srtRenderImageRanks(self.h, C.byref(params), accum.ctypes.data if want_accum else None,
                                            rgba.ctypes.data if want_rgba else None))
        return accum, rgba

    def comm_destroy(self):
        self._check(lib.srtCommDestroy(self.h))

    def trace(self, rays, traversal=abi.SRT_TRAVERSE_FAITHFUL):
        rays = np.ascontiguousarray(rays, abi.RAY_DTYPE)
        hits = np.zeros(len(rays), abi.HIT_DTYPE)
        self._check(lib.srtTraceRays(self.h, rays.ctypes.data, len(rays), hits.ctypes.data, traversal))
        return hits

    def scatter_test(self, rays, hits, seed):
        rays = np.ascontiguousarray(rays, abi.RAY_DTYPE)
        hits = np.ascontiguousarray(hits, abi.HIT_DTYPE)
        out = np.zeros((len(rays), 13), np.float32)
        self._check(lib.srtScatterTest(self.h, rays.ctypes.data, hits.ctypes.data, len(rays), seed, out.ctypes.data))
        return out

    def set_tunable(self, name, value):
        """Diagnostic knobs of the work distribution / wave scheduler (include/srt_hip_test.h)."""
        self._check(lib.srtSetTunable(self.h, name.encode(), int(value)))

    def get_tunable(self, name):
        v = C.c_int32(0)
        self._check(lib.srtGetTunable(self.h, name.encode(), C.byref(v)))
        return v.value

    def render_aov(self, params, depth=0):
        """The render kernel's own traversal of the ray at bounce `depth` of every pixel's first sample
        (include/srt_hip_test.h): returns an AOV_DTYPE array (H, W)."""
        out = np.zeros((params.imageHeight, params.imageWidth), abi.AOV_DTYPE)
        self._check(lib.srtRenderAov(self.h, C.byref(params), depth, out.ctypes.data))
        return out

    def shade_profile(self):
        out = np.zeros(10, np.uint64)
        self._check(lib.srtGetShadeProfile(self.h, out.ctypes.data))
        return [int(x) for x in out]

    def wf_profile(self):
        """Step profile of the last path-pool launch with tunable wf_profile = 1 (include/srt_hip_test.h)."""
        out = np.zeros(46, np.uint64)
        self._check(lib.srtGetWfProfile(self.h, out.ctypes.data))
        kinds = ["node", "prim", "swap", "hit0", "hit1", "hit2", "restart", "idle", "lost_claim", "new_item", "far_node", "unused"]
        K = len(kinds)
        prof = {k: {"clocks": int(out[i]), "runs": int(out[K + i]), "lanes": int(out[2 * K + i])} for i, k in enumerate(kinds)}
        prof["sched_clocks"], prof["total_clocks"] = int(out[3 * K]), int(out[3 * K + 1])
        n = max(1, int(out[3 * K + 2]))
        prof["decisions"] = int(out[3 * K + 2])
        prof["mean_seen"] = {k: float(out[3 * K + 3 + i]) / n for i, k in enumerate(["at_node", "at_prim", "finished", "idle", "ready_fill", "fullest_ring", "restart_fill"])}
        return prof

    def launch_info(self):
        """The most recent render launch (include/srt_hip_test.h).  lds_tree_mode: 0 node records through the L1,
        1 / 2 LDS-resident tree with the attenuation stacks in global memory / LDS, 3 the path-pool kernel, 4 its hybrid form
        (the tree's top in LDS, the rest read from global memory)."""
        out = np.zeros(4, np.int32)
        self._check(lib.srtGetLaunchInfo(self.h, out.ctypes.data))
        return {"lds_tree": bool(out[0]), "lds_tree_mode": int(out[0]), "wavefront": int(out[0]) in (3, 4), "hybrid": int(out[0]) == 4, "workgroups": int(out[1]),
                "threads": int(out[2]), "lds_bytes": int(out[3])}

    def last_kernel_ms(self):
        ms = C.c_float(0)
        self._check(lib.srtLastKernelMs(self.h, C.byref(ms)))
        return ms.value

    def stats(self):
        s = abi.SrtStats()
        self._check(lib.srtGetStats(self.h, C.byref(s)))
        return {n: int(getattr(s, n)) for n, _ in s._fields_}

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, mhz = C.c_int32(0), C.c_int32(0)
        self._check(lib.srtDeviceInfo(self.h, name, 256, C.byref(cus), C.byref(mhz)))
        return {"name": name.value.decode(), "cus": cus.value, "clock_mhz": mhz.value}
