/*
 * oracle.h — C API of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker.
 * The product path (pathed_amd/, libpathed_hip.so, libpathed_host.so) never
 * includes, links or calls anything under oracle/.
 *
 * It is a plain scalar restatement of the reference's radiance loop
 * (chellmuth/pathed src/path_tracer.cpp, src/sample_integrator.cpp and the
 * Scene / Material / Light / Shape code they call) with its own BVH standing in
 * for Embree.  See oracle.cpp for the per-function reference citations and for
 * what is and is not pinned against the reference.
 */
#ifndef PATHED_ORACLE_H
#define PATHED_ORACLE_H

#include "pathed_hip.h" /* scene description structs only */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleScene OracleScene;

OracleScene *oracle_scene_create(const PathedSceneDesc *desc);
void oracle_scene_destroy(OracleScene *scene);
const char *oracle_last_error(void);

/* Same contract as pathed_hip_render: adds the radiance sums of samples
 * [spp_begin, spp_begin+spp_count) of every pixel into accum_rgb_sum in sample
 * order.  threads <= 1 runs serially; otherwise OpenMP over rows like the
 * reference (src/sample_integrator.cpp:99).  stats (optional, 8 x uint64):
 * camera samples, closest rays, shadow rays, box tests, triangle tests,
 * dropped samples, path vertices, shadow rays whose light sample has a non-black unoccluded contribution. */
int oracle_render(OracleScene *scene, uint64_t seed,
                  uint32_t spp_begin, uint32_t spp_count,
                  int start_bounce, int last_bounce,
                  float *accum_rgb_sum, int threads, uint64_t *stats);

/* chunk > 1 mirrors pathed_hip_set_samples_per_unit: group sums of `chunk` samples. */
int oracle_render_chunked(OracleScene *scene, uint64_t seed,
                          uint32_t spp_begin, uint32_t spp_count,
                          int start_bounce, int last_bounce,
                          float *accum_rgb_sum, int threads, uint64_t *stats, int chunk);

/* Environment-light function-level checks (need an image, hence a scene):
 * fn in {"env_emit", "env_pdf", "env_sample"}, layouts in tests/golden/README.md. */
int oracle_env_eval(OracleScene *scene, const char *fn, const float *in, int n_in, float *out, int n_out);

/* Radiance of ONE camera sample (for spot checks): rgb out. */
int oracle_sample_pixel(OracleScene *scene, uint64_t seed,
                        int row, int col, uint32_t sample,
                        int start_bounce, int last_bounce, float *rgb);

/* Same contract as pathed_hip_trace (rays: 8 floats each). */
int oracle_trace(OracleScene *scene, const float *rays, size_t n, int any_hit, void *hits);

/* Brute-force double-precision reference intersector over the same triangles and
 * spheres: pins the oracle's own BVH + fp32 intersection. */
int oracle_trace_bruteforce(OracleScene *scene, const float *rays, size_t n, double *t_out, int32_t *prim_out);

/* Walk a BVH exported by pathed_hip_scene_export_bvh with the oracle's own
 * traversal code and count child boxes / triangles tested for the given rays
 * (the algorithmic-bytes check of DESIGN.md). counts: 2 x uint64. */
int oracle_count_exported_bvh(const float *nodes, size_t n_nodes, const float *tris, size_t n_tris,
                              const float *rays, size_t n, int any_hit, uint64_t *counts);

/* The counter-based random stream shared (by specification) with the HIP kernels. */
float oracle_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dimension);

/* Function-level evaluation for the golden-vector tests; see tests/golden/README.md
 * for the argument layout of each `fn`.  Returns the number of outputs written,
 * or a negative value for an unknown function / bad arity. */
int oracle_eval(const char *fn, const float *in, int n_in, float *out, int n_out);

int oracle_light_count(OracleScene *scene);

/* PATHED_INTEGRATOR_PATH_TRACER (default) or PATHED_INTEGRATOR_VOLUME_PATH_TRACER: which of the reference's
 * SampleIntegrator subclasses oracle_render* restate (src/path_tracer.cpp / src/volume_path_tracer.cpp). */
int oracle_set_integrator(OracleScene *scene, int integrator);

#ifdef __cplusplus
}
#endif
#endif
