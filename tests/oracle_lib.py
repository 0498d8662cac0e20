"""ctypes binding of oracle/liboracle.so — the CPU checker.

Test infrastructure: imported only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by pathed_amd/.
"""
import ctypes as C
import os

import numpy as np

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(REPO_ROOT, "oracle", "liboracle.so")

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_PATH):
        raise RuntimeError("%s missing: run `make oracle`" % ORACLE_PATH)
    lib = C.CDLL(ORACLE_PATH)
    vp = C.c_void_p
    fp = C.POINTER(C.c_float)
    lib.oracle_scene_create.argtypes = [vp]
    lib.oracle_scene_create.restype = vp
    lib.oracle_scene_destroy.argtypes = [vp]
    lib.oracle_scene_destroy.restype = None
    lib.oracle_last_error.restype = C.c_char_p
    lib.oracle_render.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, fp, C.c_int, C.POINTER(C.c_uint64)]
    lib.oracle_render.restype = C.c_int
    lib.oracle_render_chunked.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, fp, C.c_int, C.POINTER(C.c_uint64), C.c_int]
    lib.oracle_render_chunked.restype = C.c_int
    lib.oracle_sample_pixel.argtypes = [vp, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, fp]
    lib.oracle_sample_pixel.restype = C.c_int
    lib.oracle_trace.argtypes = [vp, fp, C.c_size_t, C.c_int, vp]
    lib.oracle_trace.restype = C.c_int
    lib.oracle_trace_bruteforce.argtypes = [vp, fp, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    lib.oracle_trace_bruteforce.restype = C.c_int
    lib.oracle_count_exported_bvh.argtypes = [fp, C.c_size_t, fp, C.c_size_t, fp, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)]
    lib.oracle_count_exported_bvh.restype = C.c_int
    lib.oracle_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.oracle_rng.restype = C.c_float
    lib.oracle_eval.argtypes = [C.c_char_p, fp, C.c_int, fp, C.c_int]
    lib.oracle_eval.restype = C.c_int
    lib.oracle_env_eval.argtypes = [vp, C.c_char_p, fp, C.c_int, fp, C.c_int]
    lib.oracle_env_eval.restype = C.c_int
    lib.oracle_light_count.argtypes = [vp]
    lib.oracle_light_count.restype = C.c_int
    lib.oracle_set_integrator.argtypes = [vp, C.c_int]
    lib.oracle_set_integrator.restype = C.c_int
    _lib = lib
    return lib


_threads = None


def host_threads():
    """Threads worth starting: the scheduler affinity cut to the cgroup's CPU quota and, where neither says less, to 32 -- a GPU
    box reports 256 hardware threads and lets a job run on a 16-core share of them (bench.py measures which count is
    fastest for its baseline; tests only need a sane one)."""
    global _threads
    if _threads is not None:
        return _threads
    try:
        count = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        count = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as handle:
            first, period = handle.read().split()[:2]
            if first != "max":
                count = max(1, min(count, int(float(first) / float(period) + 0.999)))
    except (OSError, ValueError):
        pass
    override = os.environ.get("PATHED_ORACLE_THREADS")
    _threads = int(override) if override else min(count, 32)
    return _threads


def _fptr(array):
    return array.ctypes.data_as(C.POINTER(C.c_float))


class OracleScene:
    def __init__(self, desc_pointer):
        self.lib = load()
        self.handle = self.lib.oracle_scene_create(C.cast(desc_pointer, C.c_void_p))
        if not self.handle:
            raise RuntimeError("oracle_scene_create: %s" % self.lib.oracle_last_error().decode())

    def close(self):
        if self.handle:
            self.lib.oracle_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def render(self, width, height, seed, spp_begin, spp_count, start_bounce, last_bounce, threads=1, accum=None,
               chunk=1):
        if accum is None:
            accum = np.zeros((height, width, 3), dtype=np.float32)
        stats = (C.c_uint64 * 8)()
        threads = max(1, min(int(threads), host_threads()))   # callers pass os.cpu_count(): never more than the job may run
        code = self.lib.oracle_render_chunked(self.handle, seed, spp_begin, spp_count, start_bounce, last_bounce,
                                              _fptr(accum), threads, stats, chunk)
        if code != 0:
            raise RuntimeError("oracle_render failed")
        names = ["camera_samples", "closest_rays", "shadow_rays", "box_tests", "tri_tests", "dropped", "vertices", "shadow_rays_needed"]
        return accum, dict(zip(names, list(stats)))

    def sample_pixel(self, seed, row, col, sample, start_bounce, last_bounce):
        rgb = np.zeros(3, dtype=np.float32)
        self.lib.oracle_sample_pixel(self.handle, seed, row, col, sample, start_bounce, last_bounce, _fptr(rgb))
        return rgb

    def trace(self, rays, any_hit=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        if any_hit:
            out = np.zeros(n, dtype=np.int32)
        else:
            out = np.zeros((n, 4), dtype=np.float32)
        self.lib.oracle_trace(self.handle, _fptr(rays), n, 1 if any_hit else 0, out.ctypes.data_as(C.c_void_p))
        return out

    def trace_bruteforce(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        t = np.zeros(n, dtype=np.float64)
        prim = np.zeros(n, dtype=np.int32)
        self.lib.oracle_trace_bruteforce(self.handle, _fptr(rays), n, t.ctypes.data_as(C.POINTER(C.c_double)),
                                         prim.ctypes.data_as(C.POINTER(C.c_int32)))
        return t, prim

    def env_eval(self, fn, inputs, n_out):
        inputs = np.ascontiguousarray(inputs, dtype=np.float32)
        out = np.zeros(n_out, dtype=np.float32)
        n = self.lib.oracle_env_eval(self.handle, fn.encode(), _fptr(inputs), inputs.size, _fptr(out), n_out)
        if n < 0:
            raise RuntimeError("oracle_env_eval(%s) -> %d" % (fn, n))
        return out[:n]

    def set_integrator(self, name):
        code = {"PathTracer": 0, "VolumePathTracer": 1}[name]
        if self.lib.oracle_set_integrator(self.handle, code) != 0:
            raise RuntimeError("oracle_set_integrator")

    def light_count(self):
        return self.lib.oracle_light_count(self.handle)


def evaluate(fn, inputs, n_out=16):
    lib = load()
    inputs = np.ascontiguousarray(inputs, dtype=np.float32)
    out = np.zeros(n_out, dtype=np.float32)
    n = lib.oracle_eval(fn.encode(), _fptr(inputs), inputs.size, _fptr(out), n_out)
    if n < 0:
        raise RuntimeError("oracle_eval(%s) -> %d" % (fn, n))
    return out[:n]


def rng(seed, pixel, sample, dimension):
    return load().oracle_rng(seed, pixel, sample, dimension)
