"""Python mirror of the reference's Integrator plug-in surface for the GPU path.

Reference: include/integrator.h:16-55 (`Integrator::run(image, scene, callback, quit)`),
src/integrator.cpp:19-106 (wave loop, sum -> mean, power-of-two checkpoints) and
src/job.cpp:65-97 (the "integrator" string factory).  The C++ twin of this file is
pathed_amd/host/integrator.{h,cpp}; both sit directly on include/pathed_hip.h.

Everything computed here happens inside libpathed_hip.so; if that library (or a GPU)
is missing the calls raise — there is no CPU fallback in the product path.
"""
import ctypes as C

import numpy as np

from . import _capi


class PathedError(RuntimeError):
    pass


def _check(lib, code, what):
    if code != 0:
        raise PathedError("%s failed (%d): %s" % (what, code, lib.pathed_hip_last_error().decode()))


def measure_bandwidth(gib=2.0, repeats=10):
    """(read GB/s, copy GB/s) of a plain streaming kernel on the current device (pathed_hip_measure_bandwidth)."""
    lib = _capi.load_hip()
    read, copy = C.c_double(0.0), C.c_double(0.0)
    _check(lib, lib.pathed_hip_measure_bandwidth(int(gib * (1 << 30)), int(repeats), C.byref(read), C.byref(copy)),
           "pathed_hip_measure_bandwidth")
    return read.value, copy.value


def measure_valu(waves_per_simd=4, repeats=10):
    """(v_fma_f32, fma + rcp/sqrt mix) wave-instructions per second the device sustains (pathed_hip_measure_valu)."""
    lib = _capi.load_hip()
    fma, mixed = C.c_double(0.0), C.c_double(0.0)
    _check(lib, lib.pathed_hip_measure_valu(int(waves_per_simd), int(repeats), C.byref(fma), C.byref(mixed)),
           "pathed_hip_measure_valu")
    return fma.value, mixed.value


VALU_MODES = ("v_fma_f32 (1 VGPR source)", "6 v_fma_f32 + v_rcp_f32 + v_sqrt_f32", "v_fma_f32 (3 VGPR sources)", "v_pk_fma_f32", "v_mul_lo_u32")


def measure_valu_modes(waves_per_simd=4, repeats=5):
    """wave-instructions per second for each of VALU_MODES (pathed_hip_measure_valu_modes)."""
    lib = _capi.load_hip()
    rates = (C.c_double * 5)()
    _check(lib, lib.pathed_hip_measure_valu_modes(int(waves_per_simd), int(repeats), rates, 5), "pathed_hip_measure_valu_modes")
    return list(rates)


def measure_valu_clocks(waves_per_simd=4, chains=8, repeats=5):
    """v_fma_f32 issue rate with the probe's own clocks (pathed_hip_measure_valu_clocks): a dict with the rate, the shader
    clock the chip ran at, and cycles per instruction."""
    lib = _capi.load_hip()
    out = _capi.PathedValuClocks()
    _check(lib, lib.pathed_hip_measure_valu_clocks(int(waves_per_simd), int(chains), int(repeats), C.byref(out)), "pathed_hip_measure_valu_clocks")
    return {name: getattr(out, name) for name, _ in _capi.PathedValuClocks._fields_}


class HipScene:
    """A scene uploaded to one GPU (PathedScene handle).

    Keyword options map onto PathedSceneOptions (include/pathed_hip.h): stack_rows, pools,
    suspend_lanes, suspend_patience, park_min_cards, max_slots, build_threads, generic_kernels, refittable, wave_max_ksamples,
    wave_stragglers, wave_refill, chunks_per_pass, hybrid_batch, hybrid_ready, local_rays, shade_launches, shade_chain, intersector ("auto" | "bvh"),
    trace_blocks_per_cu, shade_kernel ("auto" | "per-slot" | "staged" | "fused" | "split" | "wave" | "hybrid"), stage_slots,
    unit_order ("auto" | "stripes" | "stripes-tiled" | "tiles"), node_format ("auto" | "wide" | "compressed" | "compressed8"), small_phase1 ("auto" | "valu" | "mfma").  `device=None` keeps the device of an earlier pathed_hip_init.
    """

    BVH_BUILDERS = {"sah": 0, "lbvh": 1, "ploc": 2}  # PATHED_BVH_SAH_HOST / _LBVH_DEVICE / _PLOC_DEVICE

    def __init__(self, desc_pointer, device=None, bvh_builder="sah", **options):
        self._lib = _capi.load_hip()
        packed = _capi.PathedSceneOptions()
        packed.struct_size = C.sizeof(_capi.PathedSceneOptions)
        packed.device = _capi.DEVICE_CURRENT if device is None else int(device)
        packed.bvh_builder = self.BVH_BUILDERS[bvh_builder] + 1
        intersector = options.pop("intersector", "auto")
        packed.intersector = {"auto": 0, "bvh": 1}[intersector]
        packed.shade_kernel = {"auto": 0, "per-slot": 1, "staged": 2, "fused": 3, "split": 4, "wave": 5, "hybrid": 6}[options.pop("shade_kernel", "auto")]
        packed.node_format = {"auto": 0, "wide": 1, "compressed": 2, "compressed8": 3}[options.pop("node_format", "auto")]
        packed.unit_order = {"auto": 0, "stripes": 1, "stripes-tiled": 2, "tiles": 3}[options.pop("unit_order", "auto")]
        packed.small_phase1 = {"auto": 0, "valu": 1, "mfma": 2}[options.pop("small_phase1", "auto")]
        for name in ("stack_rows", "pools", "suspend_lanes", "suspend_patience", "park_min_cards", "max_slots", "trace_blocks_per_cu", "stage_slots", "build_threads", "generic_kernels", "refittable",
                     "wave_max_ksamples", "wave_stragglers", "wave_refill", "chunks_per_pass", "hybrid_batch", "hybrid_ready", "local_rays", "shade_launches", "shade_chain"):
            if name in options:
                setattr(packed, name, int(options.pop(name)))
        if options:
            raise TypeError("unknown scene options: %s" % sorted(options))
        handle = C.c_void_p()
        _check(self._lib, self._lib.pathed_hip_scene_create_ex(desc_pointer, C.byref(packed), C.byref(handle)),
               "pathed_hip_scene_create_ex")
        self._handle = handle
        self.device = int(self._lib.pathed_hip_scene_device(handle))
        self.width = int(desc_pointer.contents.camera.width)
        self.height = int(desc_pointer.contents.camera.height)

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.pathed_hip_scene_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, seed, spp_begin, spp_count, start_bounce, last_bounce, accum=None):
        """radianceLookup += ... for samples [spp_begin, spp_begin+spp_count); host buffer (H, W, 3)."""
        if accum is None:
            accum = np.zeros((self.height, self.width, 3), dtype=np.float32)
        assert accum.dtype == np.float32 and accum.flags["C_CONTIGUOUS"] and accum.size == 3 * self.width * self.height
        code = self._lib.pathed_hip_render(
            self._handle, C.c_uint64(seed), spp_begin, spp_count, start_bounce, last_bounce,
            accum.ctypes.data_as(C.POINTER(C.c_float)))
        _check(self._lib, code, "pathed_hip_render")
        return accum

    def render_device(self, seed, spp_begin, spp_count, start_bounce, last_bounce, device_pointer, stream=0):
        """Same, into caller-owned device memory (e.g. tensor.data_ptr()) on `stream`."""
        code = self._lib.pathed_hip_render_device(
            self._handle, C.c_uint64(seed), spp_begin, spp_count, start_bounce, last_bounce,
            C.c_void_p(device_pointer), C.c_void_p(stream), 1)
        _check(self._lib, code, "pathed_hip_render_device")

    def trace(self, rays, any_hit=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        out = np.zeros(n, dtype=np.int32) if any_hit else np.zeros((n, 4), dtype=np.float32)
        code = self._lib.pathed_hip_trace(
            self._handle, rays.ctypes.data_as(C.POINTER(C.c_float)), n, 1 if any_hit else 0,
            out.ctypes.data_as(C.c_void_p))
        _check(self._lib, code, "pathed_hip_trace")
        return out

    def small_candidates(self, rays):
        """Phase-1 candidate sets of both forms and the set phase 2 accepts (pathed_hip_debug_small_candidates): rays (n, 10) =
        origin, continuation direction, shadow direction, shadow tfar -> (n, 8) uint64, bit p = primitive id p."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 10)
        out = np.zeros((rays.shape[0], 8), dtype=np.uint64)
        code = self._lib.pathed_hip_debug_small_candidates(
            self._handle, rays.ctypes.data_as(C.POINTER(C.c_float)), rays.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint64)))
        _check(self._lib, code, "pathed_hip_debug_small_candidates")
        return out

    def refit(self, positions, normals=None):
        """New vertex positions (and normals) over the same topology (pathed_hip_scene_refit; scenes created with refittable=1).
        Returns the device time of the refit kernels in milliseconds."""
        positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        normal_pointer = None
        if normals is not None:
            normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
            assert normals.shape == positions.shape
            normal_pointer = normals.ctypes.data_as(C.POINTER(C.c_float))
        ms = C.c_float(0.0)
        code = self._lib.pathed_hip_scene_refit(self._handle, positions.ctypes.data_as(C.POINTER(C.c_float)), normal_pointer,
                                                positions.shape[0], C.byref(ms))
        _check(self._lib, code, "pathed_hip_scene_refit")
        return float(ms.value)

    def set_samples_per_unit(self, samples):
        """Summation granularity (see include/pathed_hip.h); 1 = the reference's exact order."""
        _check(self._lib, self._lib.pathed_hip_set_samples_per_unit(self._handle, int(samples)),
               "pathed_hip_set_samples_per_unit")

    def set_camera(self, camera):
        """Another view of the same scene (pathed_hip_scene_set_camera): `camera` is a _capi.PathedCamera of the scene's resolution."""
        _check(self._lib, self._lib.pathed_hip_scene_set_camera(self._handle, C.byref(camera)), "pathed_hip_scene_set_camera")

    def set_integrator(self, name):
        """"PathTracer" (default) or "VolumePathTracer" (reference src/job.cpp:65-97)."""
        code = {"PathTracer": _capi.INTEGRATOR_PATH_TRACER, "DataParallelIntegrator": _capi.INTEGRATOR_PATH_TRACER,
                "VolumePathTracer": _capi.INTEGRATOR_VOLUME_PATH_TRACER}[name]
        _check(self._lib, self._lib.pathed_hip_set_integrator(self._handle, code), "pathed_hip_set_integrator")

    def set_stats_mode(self, count=False, time_kernels=False, time_sampled=False):
        """time_kernels: HIP events around every launch; time_sampled: around every 8th (cheaper, same averages)."""
        mode = (1 if count else 0) | (2 if time_kernels else 0) | (4 if time_sampled else 0)
        _check(self._lib, self._lib.pathed_hip_set_stats_mode(self._handle, mode), "pathed_hip_set_stats_mode")

    def reset_stats(self):
        _check(self._lib, self._lib.pathed_hip_reset_stats(self._handle), "pathed_hip_reset_stats")

    def stats(self):
        stats = _capi.PathedStats()
        _check(self._lib, self._lib.pathed_hip_get_stats(self._handle, C.byref(stats)), "pathed_hip_get_stats")
        return {name: getattr(stats, name) for name, _ in _capi.PathedStats._fields_}

    def export_bvh(self):
        n_nodes, n_tris = C.c_size_t(0), C.c_size_t(0)
        _check(self._lib, self._lib.pathed_hip_scene_export_bvh(self._handle, None, C.byref(n_nodes), None, C.byref(n_tris)),
               "pathed_hip_scene_export_bvh")
        nodes = np.zeros((n_nodes.value, 32), dtype=np.float32)
        tris = np.zeros((n_tris.value, 12), dtype=np.float32)
        _check(self._lib, self._lib.pathed_hip_scene_export_bvh(
            self._handle, nodes.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n_nodes),
            tris.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n_tris)), "pathed_hip_scene_export_bvh")
        return nodes, tris

    def export_compressed_nodes(self):
        """(n, 16) or (n, 32) uint32 words of the compressed nodes (include/pathed_hip.h), n = 0 when the scene carries none"""
        n_nodes, words = C.c_size_t(0), C.c_size_t(0)
        _check(self._lib, self._lib.pathed_hip_scene_export_compressed_nodes(self._handle, None, C.byref(n_nodes), C.byref(words)),
               "pathed_hip_scene_export_compressed_nodes")
        nodes = np.zeros((n_nodes.value, words.value or 16), dtype=np.uint32)
        if n_nodes.value:
            _check(self._lib, self._lib.pathed_hip_scene_export_compressed_nodes(
                self._handle, nodes.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n_nodes), C.byref(words)), "pathed_hip_scene_export_compressed_nodes")
        return nodes


class BounceController:
    """reference src/bounce_controller.cpp:5-25"""

    def __init__(self, start_bounce, last_bounce):
        assert start_bounce >= 0
        assert last_bounce == -1 or start_bounce <= last_bounce
        self.start_bounce = start_bounce
        self.last_bounce = last_bounce

    def check_done(self, bounce):
        if self.last_bounce == -1:
            return False
        return bounce > self.last_bounce

    def check_counts(self, bounce):
        if self.start_bounce > bounce:
            return False
        return not self.check_done(bounce)


class PathTracer:
    """GPU stand-in for the reference's PathTracer integrator.

    `run(image, scene, callback, quit)` keeps the reference's contract: `image` receives
    sum/(i+1) after every batch, checkpoints are reported at power-of-two sample counts.
    Instead of one wave per call it renders `spp_per_launch` samples per launch (the
    reference's PDFIntegrator also overrides run()).
    """

    def __init__(self, bounce_controller, spp=1, seed=1, spp_per_launch=1024):
        self.bounce_controller = bounce_controller
        self.spp = spp
        self.seed = seed
        if int(spp_per_launch) < 1:
            raise PathedError("spp_per_launch must be >= 1")
        self.spp_per_launch = int(spp_per_launch)

    def run(self, image, scene, callback=None, quit_flag=None):
        """image: float32 (H, W, 3) array that receives the running mean; scene: HipScene."""
        radiance_lookup = np.zeros((scene.height, scene.width, 3), dtype=np.float32)
        done = 0
        while done < self.spp:
            # stop at the next power of two so checkpoints land exactly where the reference writes them
            next_power = 1
            while next_power <= done:
                next_power *= 2
            count = min(self.spp_per_launch, self.spp - done, next_power - done)
            scene.render(self.seed, done, count, self.bounce_controller.start_bounce,
                         self.bounce_controller.last_bounce, radiance_lookup)
            done += count
            np.divide(radiance_lookup, np.float32(done), out=image)
            if callback is not None:
                callback(done, (done & (done - 1)) == 0)
            if quit_flag is not None and quit_flag():
                return
        return image


def integrator_from_job(job, **kwargs):
    """reference Job::integrator(), src/job.cpp:65-97 — only the hot-path integrator exists here."""
    name = job["integrator"]
    if name in ("PathTracer", "DataParallelIntegrator", "VolumePathTracer"):
        # the caller selects the arithmetic on the scene: HipScene.set_integrator(name)
        return PathTracer(BounceController(job["startBounce"], job["lastBounce"]),
                          spp=job["spp"] if job["spp"] > 0 else 9999999, **kwargs)
    raise PathedError("Unimplemented")
