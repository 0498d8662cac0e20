"""ctypes view of include/pathed_hip.h and of the host library's loader entry points.

Plumbing only: the structs mirror the C header field for field; nothing here computes.
The HIP library is mandatory for the product path — `load_hip()` raises if it is
missing instead of falling back to anything.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
LIB_DIR = os.path.join(_HERE, "lib")

PATHED_ABI_VERSION = 4

MAT_LAMBERTIAN, MAT_OREN_NAYAR, MAT_MICROFACET, MAT_PLASTIC, MAT_GLASS, MAT_MIRROR, MAT_PASSTHROUGH = range(7)
INTEGRATOR_PATH_TRACER, INTEGRATOR_VOLUME_PATH_TRACER = 0, 1
ALBEDO_CONSTANT, ALBEDO_CHECKERBOARD, ALBEDO_TEXTURE = 0, 1, 2
GEOM_MESH, GEOM_SPHERE = 0, 1


class PathedCamera(C.Structure):
    _fields_ = [
        ("origin", C.c_float * 3),
        ("target", C.c_float * 3),
        ("up", C.c_float * 3),
        ("vertical_fov", C.c_float),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("flip_handedness", C.c_int32),
    ]


class PathedMaterial(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("albedo_type", C.c_int32),
        ("diffuse", C.c_float * 3),
        ("emit", C.c_float * 3),
        ("checker_on", C.c_float * 3),
        ("checker_off", C.c_float * 3),
        ("checker_res", C.c_float * 2),
        ("sigma", C.c_float),
        ("alpha", C.c_float),
        ("ior", C.c_float),
        ("distribution", C.c_int32),
        ("texture", C.c_int32),
    ]


class PathedSphere(C.Structure):
    _fields_ = [
        ("center_world", C.c_float * 3),
        ("radius", C.c_float),
        ("center_sample", C.c_float * 3),
        ("material", C.c_int32),
    ]


class PathedGeom(C.Structure):
    _fields_ = [("type", C.c_int32), ("first", C.c_int32), ("count", C.c_int32), ("medium", C.c_int32)]


class PathedMedium(C.Structure):
    _fields_ = [("sigma_t", C.c_float * 3), ("sigma_s", C.c_float * 3)]


class PathedEnvLight(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("rgba", C.POINTER(C.c_float)),
        ("scale", C.c_float),
        ("map_to_world", C.c_float * 16),
        ("world_to_map", C.c_float * 16),
    ]


class PathedTexture(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("rgb", C.POINTER(C.c_uint8)),
    ]


class PathedSceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("camera", PathedCamera),
        ("n_vertices", C.c_uint32),
        ("positions", C.POINTER(C.c_float)),
        ("normals", C.POINTER(C.c_float)),
        ("uvs", C.POINTER(C.c_float)),
        ("n_triangles", C.c_uint32),
        ("indices", C.POINTER(C.c_uint32)),
        ("tri_material", C.POINTER(C.c_int32)),
        ("n_spheres", C.c_uint32),
        ("spheres", C.POINTER(PathedSphere)),
        ("n_geoms", C.c_uint32),
        ("geoms", C.POINTER(PathedGeom)),
        ("n_materials", C.c_uint32),
        ("materials", C.POINTER(PathedMaterial)),
        ("env", C.POINTER(PathedEnvLight)),
        ("n_textures", C.c_uint32),
        ("textures", C.POINTER(PathedTexture)),
        ("n_media", C.c_uint32),
        ("media", C.POINTER(PathedMedium)),
    ]


class PathedStats(C.Structure):
    _fields_ = [
        ("camera_samples", C.c_uint64),
        ("closest_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("nodes_visited", C.c_uint64),
        ("tris_tested", C.c_uint64),
        ("dropped_samples", C.c_uint64),
        ("iterations", C.c_uint64),
        ("trace_ms", C.c_double),
        ("shade_ms", C.c_double),
        ("trace_launches", C.c_uint64),
        ("bvh_nodes", C.c_uint64),
        ("bvh_bytes", C.c_uint64),
        ("bvh_max_depth", C.c_uint32),
        ("scene_in_lds", C.c_uint32),
        ("max_boxes_per_ray", C.c_uint64),
        ("parked_rays", C.c_uint64),
        ("bvh_build_ms", C.c_double),
        ("bvh_builder", C.c_uint32),
        ("trace_launches_all", C.c_uint32),
        ("path_kernel", C.c_uint32),
        ("reserved0", C.c_uint32),
        ("local_closest_rays", C.c_uint64),
        ("local_shadow_rays", C.c_uint64),
    ]


class PathedSceneOptions(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("bvh_builder", C.c_int32),
        ("stack_rows", C.c_int32),
        ("pools", C.c_int32),
        ("suspend_lanes", C.c_int32),
        ("suspend_patience", C.c_int32),
        ("park_min_cards", C.c_int32),
        ("max_slots", C.c_int32),
        ("intersector", C.c_int32),
        ("trace_blocks_per_cu", C.c_int32),
        ("shade_kernel", C.c_int32),
        ("stage_slots", C.c_int32),
        ("unit_order", C.c_int32),
        ("build_threads", C.c_int32),
        ("generic_kernels", C.c_int32),
        ("node_format", C.c_int32),
        ("small_phase1", C.c_int32),
        ("refittable", C.c_int32),
        ("wave_max_ksamples", C.c_int32),
        ("wave_stragglers", C.c_int32),
        ("wave_refill", C.c_int32),
        ("chunks_per_pass", C.c_int32),
        ("local_rays", C.c_int32),
        ("shade_chain", C.c_int32),
        ("shade_launches", C.c_int32),
        ("hybrid_batch", C.c_int32),
        ("hybrid_ready", C.c_int32),
    ]


class PathedValuClocks(C.Structure):
    _fields_ = [(name, C.c_double) for name in (
        "rate", "shader_clock_mhz", "wall_clock_mhz", "peak_clock_mhz", "wave_ticks_per_instruction",
        "cycles_per_instruction", "cycles_per_instruction_events")]


DEVICE_CURRENT = -1

# every symbol include/pathed_hip.h declares; tests check that the library exports all
HIP_SYMBOLS = [
    "pathed_hip_init",
    "pathed_hip_scene_create",
    "pathed_hip_scene_create_ex",
    "pathed_hip_scene_device",
    "pathed_hip_scene_set_camera",
    "pathed_hip_scene_destroy",
    "pathed_hip_render",
    "pathed_hip_render_device",
    "pathed_hip_trace",
    "pathed_hip_debug_small_candidates",
    "pathed_hip_has_experiments",
    "pathed_hip_scene_refit",
    "pathed_hip_set_samples_per_unit",
    "pathed_hip_set_integrator",
    "pathed_hip_set_stats_mode",
    "pathed_hip_get_stats",
    "pathed_hip_reset_stats",
    "pathed_hip_scene_export_bvh",
    "pathed_hip_scene_export_compressed_nodes",
    "pathed_hip_measure_bandwidth",
    "pathed_hip_measure_valu",
    "pathed_hip_measure_valu_modes",
    "pathed_hip_measure_valu_clocks",
    "pathed_hip_accum_copy_peer",
    "pathed_hip_accum_add",
    "pathed_hip_accum_alloc",
    "pathed_hip_accum_free",
    "pathed_hip_accum_download",
    "pathed_hip_accum_upload",
    "pathed_hip_comm_init",
    "pathed_hip_comm_reduce",
    "pathed_hip_comm_destroy",
    "pathed_hip_last_error",
    "pathed_hip_version",
]

_hip = None
_host = None


class PathedLibraryMissing(RuntimeError):
    pass


def hip_library_path():
    # PATHED_HIP_LIB selects an experimental build of the same ABI (kernel tuning sweeps)
    return os.environ.get("PATHED_HIP_LIB") or os.path.join(LIB_DIR, "libpathed_hip.so")


def host_library_path():
    return os.path.join(LIB_DIR, "libpathed_host.so")


def load_hip():
    """Load libpathed_hip.so and declare its prototypes. No fallback exists."""
    global _hip
    if _hip is not None:
        return _hip
    path = hip_library_path()
    if not os.path.exists(path):
        raise PathedLibraryMissing(
            "%s not found: build it with `make hip` (python -c 'import __graft_entry__ as g; g.build()')" % path
        )
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    lib.pathed_hip_init.argtypes = [C.c_int]
    lib.pathed_hip_init.restype = C.c_int
    lib.pathed_hip_scene_create.argtypes = [C.POINTER(PathedSceneDesc), C.POINTER(vp)]
    lib.pathed_hip_scene_create.restype = C.c_int
    lib.pathed_hip_scene_create_ex.argtypes = [C.POINTER(PathedSceneDesc), C.POINTER(PathedSceneOptions), C.POINTER(vp)]
    lib.pathed_hip_scene_create_ex.restype = C.c_int
    lib.pathed_hip_scene_device.argtypes = [vp]
    lib.pathed_hip_scene_device.restype = C.c_int
    lib.pathed_hip_scene_set_camera.argtypes = [vp, C.POINTER(PathedCamera)]
    lib.pathed_hip_scene_set_camera.restype = C.c_int
    lib.pathed_hip_measure_valu.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.pathed_hip_measure_valu.restype = C.c_int
    lib.pathed_hip_measure_valu_modes.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
    lib.pathed_hip_measure_valu_modes.restype = C.c_int
    lib.pathed_hip_measure_valu_clocks.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(PathedValuClocks)]
    lib.pathed_hip_measure_valu_clocks.restype = C.c_int
    fp = C.POINTER(C.c_float)
    lib.pathed_hip_accum_copy_peer.argtypes = [vp, vp, vp, vp, C.c_size_t]
    lib.pathed_hip_accum_copy_peer.restype = C.c_int
    lib.pathed_hip_accum_add.argtypes = [vp, vp, vp, C.c_size_t]
    lib.pathed_hip_accum_add.restype = C.c_int
    lib.pathed_hip_accum_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.pathed_hip_accum_alloc.restype = C.c_int
    lib.pathed_hip_accum_free.argtypes = [vp, vp]
    lib.pathed_hip_accum_free.restype = C.c_int
    lib.pathed_hip_accum_download.argtypes = [vp, vp, C.c_size_t, fp]
    lib.pathed_hip_accum_download.restype = C.c_int
    lib.pathed_hip_accum_upload.argtypes = [vp, vp, C.c_size_t, fp]
    lib.pathed_hip_accum_upload.restype = C.c_int
    lib.pathed_hip_comm_init.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    lib.pathed_hip_comm_init.restype = C.c_int
    lib.pathed_hip_comm_reduce.argtypes = [vp, C.POINTER(vp), vp, C.c_size_t]
    lib.pathed_hip_comm_reduce.restype = C.c_int
    lib.pathed_hip_comm_destroy.argtypes = [vp]
    lib.pathed_hip_comm_destroy.restype = None
    lib.pathed_hip_scene_destroy.argtypes = [vp]
    lib.pathed_hip_scene_destroy.restype = None
    lib.pathed_hip_render.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.pathed_hip_render.restype = C.c_int
    lib.pathed_hip_render_device.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp, vp, C.c_int]
    lib.pathed_hip_render_device.restype = C.c_int
    lib.pathed_hip_trace.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, C.c_int, vp]
    lib.pathed_hip_trace.restype = C.c_int
    lib.pathed_hip_debug_small_candidates.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_uint64)]
    lib.pathed_hip_debug_small_candidates.restype = C.c_int
    lib.pathed_hip_scene_refit.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_float)]
    lib.pathed_hip_scene_refit.restype = C.c_int
    lib.pathed_hip_has_experiments.argtypes = []
    lib.pathed_hip_has_experiments.restype = C.c_int
    lib.pathed_hip_set_samples_per_unit.argtypes = [vp, C.c_int]
    lib.pathed_hip_set_samples_per_unit.restype = C.c_int
    lib.pathed_hip_set_integrator.argtypes = [vp, C.c_int]
    lib.pathed_hip_set_integrator.restype = C.c_int
    lib.pathed_hip_set_stats_mode.argtypes = [vp, C.c_int]
    lib.pathed_hip_set_stats_mode.restype = C.c_int
    lib.pathed_hip_get_stats.argtypes = [vp, C.POINTER(PathedStats)]
    lib.pathed_hip_get_stats.restype = C.c_int
    lib.pathed_hip_reset_stats.argtypes = [vp]
    lib.pathed_hip_reset_stats.restype = C.c_int
    lib.pathed_hip_scene_export_bvh.argtypes = [
        vp, C.POINTER(C.c_float), C.POINTER(C.c_size_t), C.POINTER(C.c_float), C.POINTER(C.c_size_t)
    ]
    lib.pathed_hip_scene_export_bvh.restype = C.c_int
    lib.pathed_hip_scene_export_compressed_nodes.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.pathed_hip_scene_export_compressed_nodes.restype = C.c_int
    lib.pathed_hip_measure_bandwidth.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.pathed_hip_measure_bandwidth.restype = C.c_int
    lib.pathed_hip_last_error.argtypes = []
    lib.pathed_hip_last_error.restype = C.c_char_p
    lib.pathed_hip_version.argtypes = []
    lib.pathed_hip_version.restype = C.c_char_p
    _hip = lib
    return lib


def load_host():
    """Load libpathed_host.so (scene readers, job runner)."""
    global _host
    if _host is not None:
        return _host
    path = host_library_path()
    if not os.path.exists(path):
        raise PathedLibraryMissing("%s not found: build it with `make host`" % path)
    # libpathed_host.so links against libpathed_hip.so; make sure it resolves in-tree
    if os.path.exists(hip_library_path()):
        load_hip()
    lib = C.CDLL(path)
    lib.pathed_host_load_scene.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p]
    lib.pathed_host_load_scene.restype = C.c_void_p
    lib.pathed_host_scene_desc.argtypes = [C.c_void_p]
    lib.pathed_host_scene_desc.restype = C.POINTER(PathedSceneDesc)
    lib.pathed_host_free_scene.argtypes = [C.c_void_p]
    lib.pathed_host_free_scene.restype = None
    lib.pathed_host_last_error.argtypes = []
    lib.pathed_host_last_error.restype = C.c_char_p
    if hasattr(lib, "pathed_host_run_job"):
        lib.pathed_host_run_job.argtypes = [C.c_char_p, C.c_char_p]
        lib.pathed_host_run_job.restype = C.c_int
    if hasattr(lib, "pathed_host_write_exr_float_rgba"):
        lib.pathed_host_write_exr_float_rgba.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
        lib.pathed_host_write_exr_float_rgba.restype = C.c_int
    if hasattr(lib, "pathed_host_write_exr_half_bgr"):
        lib.pathed_host_write_exr_half_bgr.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
        lib.pathed_host_write_exr_half_bgr.restype = C.c_int
    if hasattr(lib, "pathed_host_write_bmp_rgb8"):
        lib.pathed_host_write_bmp_rgb8.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
        lib.pathed_host_write_bmp_rgb8.restype = C.c_int
    if hasattr(lib, "pathed_host_load_image_rgb8"):
        lib.pathed_host_load_image_rgb8.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint8), C.c_size_t]
        lib.pathed_host_load_image_rgb8.restype = C.c_int
    if hasattr(lib, "pathed_host_read_exr_rgba"):
        lib.pathed_host_read_exr_rgba.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_size_t]
        lib.pathed_host_read_exr_rgba.restype = C.c_int
    _host = lib
    return lib
