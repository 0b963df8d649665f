"""ctypes mirror of include/srt_hip.h (the C-ABI structs) plus a small scene
builder that fills an SrtSceneDesc from numpy arrays.

Host-side plumbing only: no arithmetic of the hot path lives here."""
import ctypes as C

import numpy as np

SRT_PRIM_TRIANGLE, SRT_PRIM_SPHERE = 0, 1
SRT_MAT_PBR, SRT_MAT_METAL, SRT_MAT_DIELECTRIC, SRT_MAT_LIGHT = 0, 1, 2, 3
SRT_TEX_SOLID, SRT_TEX_CHECKER, SRT_TEX_IMAGE = 0, 1, 2
SRT_WORLD_PRIM, SRT_WORLD_BVH = 0, 1
SRT_TRAVERSE_FAITHFUL, SRT_TRAVERSE_CLOSEST = 0, 1
SRT_BUILDER_REFERENCE, SRT_BUILDER_LBVH, SRT_BUILDER_PLOC = 0, 1, 2
SRT_TILE_W = SRT_TILE_H = 8
SRT_TILE_PIXELS = 64
SRT_NO_HIT = -1

f32, i32, i64, u64 = C.c_float, C.c_int32, C.c_int64, C.c_uint64


class SrtTriangleIn(C.Structure):
    _fields_ = [("p", f32 * 3 * 3), ("uv", f32 * 2 * 3), ("material", i32)]


class SrtSphereIn(C.Structure):
    _fields_ = [("center0", f32 * 3), ("center1", f32 * 3), ("time0", f32), ("time1", f32),
                ("radius", f32), ("material", i32)]


class SrtPrimRef(C.Structure):
    _fields_ = [("type", i32), ("index", i32)]


class SrtWorldItem(C.Structure):
    _fields_ = [("kind", i32), ("first", i32), ("count", i32), ("time0", f32), ("time1", f32),
                ("numNodes", i32), ("nodes", C.c_void_p), ("builder", i32), ("pad", i32)]


class SrtMaterialIn(C.Structure):
    _fields_ = [("type", i32), ("albedoTex", i32), ("normalTex", i32), ("metallicTex", i32),
                ("roughnessTex", i32), ("albedo", f32 * 4), ("metalness", f32), ("roughness", f32),
                ("fuzz", f32), ("ir", f32), ("pad", i32 * 3)]


class SrtTextureIn(C.Structure):
    _fields_ = [("kind", i32), ("width", i32), ("height", i32), ("bpp", i32), ("texelOffset", i64),
                ("even", i32), ("odd", i32), ("color", f32 * 3), ("pad", i32)]


class SrtSceneDesc(C.Structure):
    _fields_ = [("numTriangles", i32), ("triangles", C.POINTER(SrtTriangleIn)),
                ("numSpheres", i32), ("spheres", C.POINTER(SrtSphereIn)),
                ("numPrims", i32), ("prims", C.POINTER(SrtPrimRef)),
                ("numWorld", i32), ("world", C.POINTER(SrtWorldItem)),
                ("numMaterials", i32), ("materials", C.POINTER(SrtMaterialIn)),
                ("numTextures", i32), ("textures", C.POINTER(SrtTextureIn)),
                ("numTexelBytes", i64), ("texels", C.POINTER(C.c_uint8))]


class SrtCameraParams(C.Structure):
    _fields_ = [("eye", f32 * 3), ("lookAt", f32 * 3), ("up", f32 * 3), ("vfovDegrees", f32),
                ("aspect", f32), ("aperture", f32), ("focusDist", f32), ("time0", f32), ("time1", f32)]


class SrtCamera(C.Structure):
    _fields_ = [("origin", f32 * 3), ("lleft", f32 * 3), ("horizontal", f32 * 3), ("vertical", f32 * 3),
                ("w", f32 * 3), ("hor", f32 * 3), ("vert", f32 * 3), ("lensRadius", f32),
                ("time0", f32), ("time1", f32)]


class SrtBvhNode(C.Structure):
    _fields_ = [("bmin", f32 * 3), ("left", i32), ("bmax", f32 * 3), ("right", i32)]


class SrtRay(C.Structure):
    _fields_ = [("o", f32 * 3), ("d", f32 * 3), ("time", f32), ("tMin", f32), ("tMax", f32)]


class SrtHit(C.Structure):
    _fields_ = [("prim", i32), ("t", f32), ("p", f32 * 3), ("normal", f32 * 3), ("tangent", f32 * 3),
                ("bitangent", f32 * 3), ("uv", f32 * 2), ("frontFace", i32), ("material", i32),
                ("nodeVisits", i32), ("boxPasses", i32), ("triTests", i32), ("sphereTests", i32)]


class SrtRenderParams(C.Structure):
    _fields_ = [("imageWidth", i32), ("imageHeight", i32), ("spp", i32), ("maxBounce", i32),
                ("seed", u64), ("background", f32 * 3), ("tMin", f32), ("traversal", i32),
                ("tileFirst", i32), ("tileStride", i32), ("sppChunks", i32), ("countStats", i32),
                ("sampleFirst", i32)]


class SrtStats(C.Structure):
    _fields_ = [("samples", u64), ("rays", u64), ("nodeVisits", u64), ("boxPasses", u64),
                ("triTests", u64), ("sphereTests", u64), ("shadedTriHits", u64), ("texelFetches", u64),
                ("cyclesNode", u64), ("cyclesPrim", u64), ("cyclesShade", u64), ("cyclesTotal", u64),
                ("stepsNode", u64), ("stepsPrim", u64), ("stepsShade", u64),
                ("lanesNode", u64), ("lanesPrim", u64), ("lanesShade", u64)]


RAY_DTYPE = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("time", "<f4"), ("tMin", "<f4"), ("tMax", "<f4")])
HIT_DTYPE = np.dtype([("prim", "<i4"), ("t", "<f4"), ("p", "<f4", 3), ("normal", "<f4", 3),
                      ("tangent", "<f4", 3), ("bitangent", "<f4", 3), ("uv", "<f4", 2),
                      ("frontFace", "<i4"), ("material", "<i4"), ("nodeVisits", "<i4"),
                      ("boxPasses", "<i4"), ("triTests", "<i4"), ("sphereTests", "<i4")])
AOV_DTYPE = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("time", "<f4"), ("valid", "<i4"), ("prim", "<i4"), ("t", "<f4"),
                      ("nodeVisits", "<i4"), ("boxPasses", "<i4"), ("triTests", "<i4"), ("sphereTests", "<i4"), ("pad", "<i4", 2)])
assert AOV_DTYPE.itemsize == 64
NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("left", "<i4"), ("bmax", "<f4", 3), ("right", "<i4")])
assert RAY_DTYPE.itemsize == C.sizeof(SrtRay)
assert HIT_DTYPE.itemsize == C.sizeof(SrtHit)
assert NODE_DTYPE.itemsize == C.sizeof(SrtBvhNode) == 32
assert C.sizeof(SrtTriangleIn) == 64


def default_render_params(width, height, spp, max_bounce, seed=1, background=(0.53, 0.81, 0.92),
                          traversal=SRT_TRAVERSE_FAITHFUL, tile_first=0, tile_stride=1, spp_chunks=1,
                          count_stats=0, sample_first=0):
    """main.cpp:170-180 defaults (background sky blue, tMin 0.001)."""
    p = SrtRenderParams()
    p.imageWidth, p.imageHeight, p.spp, p.maxBounce = width, height, spp, max_bounce
    p.seed = seed
    p.background[:] = background
    p.tMin = 0.001
    p.traversal = traversal
    p.tileFirst, p.tileStride, p.sppChunks, p.countStats = tile_first, tile_stride, spp_chunks, count_stats
    p.sampleFirst = sample_first
    return p


def default_camera_params(aspect=16.0 / 9.0):
    """main.cpp:164-172."""
    c = SrtCameraParams()
    c.eye[:] = (0.0, 3.0, 5.0)
    c.lookAt[:] = (0.0, 2.5, 0.0)
    c.up[:] = (0.0, 1.0, 0.0)
    c.vfovDegrees, c.aspect, c.aperture, c.focusDist, c.time0, c.time1 = 70.0, aspect, 0.1, 10.0, 0.0, 1.0
    return c


class SceneBuilder:
    """Accumulates a scene in the reference's construction vocabulary
    (textures, materials, spheres, triangles, hittableList, bvhNode) and emits
    an SrtSceneDesc whose arrays stay alive as long as this object does."""

    def __init__(self):
        self.textures, self.materials, self.triangles, self.spheres = [], [], [], []
        self._prim_chunks, self.world = [], []
        self.num_prims = 0
        self._tri_count = 0
        self.texels = bytearray()
        self._keep = None

    def _add_prims(self, ptype, first, n):
        a = np.empty((n, 2), dtype=np.int32)
        a[:, 0] = ptype
        a[:, 1] = np.arange(first, first + n, dtype=np.int32)
        self._prim_chunks.append(a)
        self.num_prims += n

    # ---- textures (texture.h)
    def solid(self, r, g, b):
        t = SrtTextureIn(kind=SRT_TEX_SOLID, even=-1, odd=-1)
        t.color[:] = (r, g, b)
        self.textures.append(t)
        return len(self.textures) - 1

    def checker(self, c_even, c_odd):
        """checker(color3f c1, color3f c2): even=c1, odd=c2 (texture.h:40)."""
        e, o = self.solid(*c_even), self.solid(*c_odd)
        self.textures.append(SrtTextureIn(kind=SRT_TEX_CHECKER, even=e, odd=o))
        return len(self.textures) - 1

    def image(self, pixels, bpp):
        """imagePNG(filename, bpp): pixels = uint8 array (h, w, bpp) as stbi_load returns it;
        None models a failed load (texture.h:117-120)."""
        t = SrtTextureIn(kind=SRT_TEX_IMAGE, even=-1, odd=-1, bpp=bpp)
        if pixels is not None:
            a = np.ascontiguousarray(pixels, dtype=np.uint8).reshape(pixels.shape[0], pixels.shape[1], bpp)
            t.height, t.width = a.shape[0], a.shape[1]
            t.texelOffset = len(self.texels)
            self.texels += a.tobytes()
        self.textures.append(t)
        return len(self.textures) - 1

    # ---- materials (material.h)
    def pbr(self, albedo_tex=-1, normal_tex=-1, metallic_tex=-1, roughness_tex=-1,
            albedo=(1.0, 1.0, 1.0, 1.0), metalness=0.0, roughness=0.0):
        m = SrtMaterialIn(type=SRT_MAT_PBR, albedoTex=albedo_tex, normalTex=normal_tex,
                          metallicTex=metallic_tex, roughnessTex=roughness_tex,
                          metalness=metalness, roughness=roughness)
        m.albedo[:] = albedo
        self.materials.append(m)
        return len(self.materials) - 1

    def metal(self, albedo, fuzz):
        m = SrtMaterialIn(type=SRT_MAT_METAL, albedoTex=-1, normalTex=-1, metallicTex=-1, roughnessTex=-1,
                          fuzz=min(fuzz, 1.0))
        m.albedo[:] = (albedo[0], albedo[1], albedo[2], 1.0)
        self.materials.append(m)
        return len(self.materials) - 1

    def dielectric(self, ir):
        self.materials.append(SrtMaterialIn(type=SRT_MAT_DIELECTRIC, albedoTex=-1, normalTex=-1,
                                            metallicTex=-1, roughnessTex=-1, ir=ir))
        return len(self.materials) - 1

    def light(self, color=None, emit_tex=None):
        """diffuseLight (material.h:138-150): a colour, or any texture id as the emit texture."""
        t = self.solid(*color) if emit_tex is None else emit_tex
        self.materials.append(SrtMaterialIn(type=SRT_MAT_LIGHT, albedoTex=t, normalTex=-1,
                                            metallicTex=-1, roughnessTex=-1))
        return len(self.materials) - 1

    # ---- primitives, appended in hittableList order
    def add_sphere(self, center, radius, material, center1=None, time0=0.0, time1=1.0):
        s = SrtSphereIn(time0=time0, time1=time1, radius=radius, material=material)
        s.center0[:] = center
        s.center1[:] = center if center1 is None else center1
        self.spheres.append(s)
        self._add_prims(SRT_PRIM_SPHERE, len(self.spheres) - 1, 1)
        return self.num_prims - 1

    def add_triangles(self, positions, texcoords, indices, material):
        """One glTF primitive (model.h:442-454): positions (n,3) f32, texcoords (n,2) f32,
        indices (m,3) integer."""
        positions = np.asarray(positions, np.float32)
        texcoords = np.asarray(texcoords, np.float32)
        indices = np.asarray(indices).reshape(-1, 3)
        n = len(indices)
        arr = np.zeros(n, dtype=np.dtype([("p", "<f4", (3, 3)), ("uv", "<f4", (3, 2)), ("material", "<i4")]))
        arr["p"] = positions[indices]
        arr["uv"] = texcoords[indices]
        arr["material"] = material
        self.triangles.append(arr)  # chunk; flattened in desc()
        self._add_prims(SRT_PRIM_TRIANGLE, self._tri_count, n)
        self._tri_count += n
        return self.num_prims - n

    # ---- world (main.cpp:146)
    def world_bvh(self, first=0, count=None, time0=0.0, time1=1.0, builder=SRT_BUILDER_REFERENCE):
        if count is None:
            count = self.num_prims - first
        self.world.append(SrtWorldItem(SRT_WORLD_BVH, first, count, time0, time1, 0, None, builder, 0))

    def world_prebuilt(self, nodes, first=0, count=None, time0=0.0, time1=1.0):
        """A caller-built tree (NODE_DTYPE array, pre-order, child refs as in SrtBvhNode)."""
        if count is None:
            count = self.num_prims - first
        nodes = np.ascontiguousarray(nodes, NODE_DTYPE)
        self._prebuilt = getattr(self, "_prebuilt", []) + [nodes]
        self.world.append(SrtWorldItem(SRT_WORLD_BVH, first, count, time0, time1, len(nodes), nodes.ctypes.data, 0, 0))

    def world_prim(self, prim):
        self.world.append(SrtWorldItem(SRT_WORLD_PRIM, prim, 1, 0.0, 0.0, 0, None, 0, 0))

    def desc(self):
        def arr(ctype, items):
            a = (ctype * max(1, len(items)))()
            for i, it in enumerate(items):
                a[i] = it
            return a

        tri_np = (np.concatenate(self.triangles) if self.triangles
                  else np.zeros(0, dtype=np.dtype([("p", "<f4", (3, 3)), ("uv", "<f4", (3, 2)), ("material", "<i4")])))
        tri_np = np.ascontiguousarray(tri_np)
        assert tri_np.dtype.itemsize == 64
        sph = arr(SrtSphereIn, self.spheres)
        prims = np.ascontiguousarray(np.concatenate(self._prim_chunks) if self._prim_chunks
                                     else np.zeros((0, 2), np.int32))
        world = arr(SrtWorldItem, self.world)
        mats = arr(SrtMaterialIn, self.materials)
        texs = arr(SrtTextureIn, self.textures)
        texels = np.frombuffer(bytes(self.texels) + b"\0" * 16, dtype=np.uint8).copy()
        d = SrtSceneDesc()
        d.numTriangles = len(tri_np)
        d.triangles = tri_np.ctypes.data_as(C.POINTER(SrtTriangleIn))
        d.numSpheres, d.spheres = len(self.spheres), sph
        d.numPrims, d.prims = len(prims), prims.ctypes.data_as(C.POINTER(SrtPrimRef))
        d.numWorld, d.world = len(self.world), world
        d.numMaterials, d.materials = len(self.materials), mats
        d.numTextures, d.textures = len(self.textures), texs
        d.numTexelBytes = len(self.texels)
        d.texels = texels.ctypes.data_as(C.POINTER(C.c_uint8))
        self._keep = (tri_np, sph, prims, world, mats, texs, texels)
        return d
