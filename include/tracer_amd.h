/*
 * tracer_amd.h -- C-ABI of the MI355X-native Monte-Carlo ray-tracing core.
 *
 * This is the drop-in boundary for ONE hot path of casselineau/Tracer:
 * TracerEngine.ray_tracer() (reference: tracer/tracer_engine.py:124-295) and
 * everything it calls per ray per bounce.  The reference has no FFI (it is pure
 * NumPy); each entry point below cites the reference Python interface it
 * replaces.  The Python host layer (the tracer_amd package) binds these through
 * ctypes; INTEGRATION.md shows the binding a Tracer maintainer would add.
 *
 * Conventions
 *   - every function returns TRC_OK (0) or a negative trc_status; the message
 *     of the last failure on the calling thread is trc_last_error().
 *   - the caller owns every host buffer; the library owns device memory.
 *   - all real data is float64 (the reference computes in float64 throughout,
 *     tracer/ray_bundle.py:25-33); indices are int64 / int32 as declared.
 *   - vectors cross the boundary as structure-of-arrays (one pointer per
 *     component) so that HBM loads are coalesced; the Python shim passes the
 *     rows of the reference's (3,N) arrays without copying.
 *   - no global state: a trc_ctx is bound to one GPU; scenes/results belong
 *     to a context.  One context per (thread, GPU).
 */
#ifndef TRACER_AMD_H
#define TRACER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRC_ABI_VERSION 2

typedef enum trc_status {
    TRC_OK = 0,
    TRC_ERR_INVALID = -1,      /* bad argument (reference: ValueError / AttributeError) */
    TRC_ERR_DEVICE = -2,       /* HIP runtime failure */
    TRC_ERR_UNSUPPORTED = -3,  /* kind not in the native table */
    TRC_ERR_CAPACITY = -4,     /* a fixed-capacity device buffer overflowed */
    TRC_ERR_NOMEM = -5
} trc_status;

/* ---- geometry managers (reference classes in parentheses) ------------------ */
typedef enum trc_gm_kind {
    TRC_GM_FLAT_INF = 0,           /* FlatGeometryManager        flat_surface.py:11-113   */
    TRC_GM_RECT = 1,               /* RectPlateGM                flat_surface.py:181-211  gm: w/2,h/2 */
    TRC_GM_RECT_EXTRUDED = 2,      /* ExtrudedRectPlateGM        flat_surface.py:253-274  gm: w/2,h/2,cx,cy,ew/2,eh/2 */
    TRC_GM_RECT_PERFORATED = 3,    /* PerforatedRectPlateGM      flat_surface.py:357-377  gm: w/2,h/2 ; extra: n*(cx,cy,r) */
    TRC_GM_ROUND = 4,              /* RoundPlateGM               flat_surface.py:457-492  gm: Re,Ri(<0:none) */
    TRC_GM_ROUND_CUT = 5,          /* StraightCutRoundPlateGM    flat_surface.py:548-560  gm: Re,Ri,x_cut */
    TRC_GM_TRIANGLE = 6,           /* TriangularFace             triangular_face.py:35-74 gm: v0xyz,v1xyz */
    TRC_GM_PARABOLOID = 7,         /* Paraboloid                 paraboloid.py:11-69      gm: a,b */
    TRC_GM_PARAB_DISH = 8,         /* ParabolicDishGM            paraboloid.py:71-119     gm: a,b,h */
    TRC_GM_PARAB_HEX = 9,          /* HexagonalParabolicDishGM   paraboloid.py:174-223    gm: a,b,R */
    TRC_GM_PARAB_RECT = 10,        /* RectangularParabolicDishGM paraboloid.py:225-294    gm: a,b,w/2,h/2 */
    TRC_GM_PARAB_RECT_OFFAXIS = 11,/* same, off_axis_normal set  paraboloid.py:279-281    gm: a,b,w/2,h/2,rot[9],centre[3] */
    TRC_GM_PARAB_CYL = 12,         /* ParabolicCylinder          paraboloid.py:328-384    gm: a */
    TRC_GM_PARAB_TROUGH = 13,      /* ParabolicTroughGM          paraboloid.py:386-443    gm: a,l/2,h */
    TRC_GM_SPHERE = 14,            /* SphericalGM                sphere_surface.py:9-68   gm: r */
    TRC_GM_HEMISPHERE = 15,        /* HemisphereGM               sphere_surface.py:117-139 gm: r */
    TRC_GM_SPHERE_RECT = 16,       /* SphericalRectFacet         sphere_surface.py:206-228 gm: r,lx/2,ly/2 */
    TRC_GM_CYL_INF = 17,           /* InfiniteCylinder           cylinder.py:12-57        gm: R */
    TRC_GM_CYL_FINITE = 18,        /* FiniteCylinder             cylinder.py:59-110       gm: R,h/2,a0,a1 */
    TRC_GM_CYL_RECTCUT = 19,       /* RectCutCylinder            cylinder.py:161-199      gm: R,h/2,w/2,hh/2 */
    TRC_GM_CONE_INF = 20,          /* InfiniteCone               cone.py:7-72             gm: c,a */
    TRC_GM_CONE_FINITE = 21,       /* FiniteCone (and RectCutCone, which the reference
                                      silently runs as FiniteCone, cone.py:161)  cone.py:74-120  gm: c,a,h */
    TRC_GM_FRUSTUM = 22,           /* ConicalFrustum             cone.py:261-320          gm: c,a,zmin,zmax */
    TRC_GM_FRUSTUM_RECTCUT = 23,   /* RectCutConicalFrustum      cone.py:356-395          gm: c,a,zmin,zmax,w/2,h/2 */
    TRC_GM_QUADRATIC = 24,         /* FlatQuadricSurfaceGM       quadratic_surface.py:4-61  gm: a,b,c,d,e,f */
    TRC_GM_QUADRATIC_RECT = 25,    /* RectFlatQuadricSurfaceGM   quadratic_surface.py:64-105 gm: a..f,w/2,h/2 */
    TRC_GM_ELLIPSOID = 26,         /* Ellipsoid                  ellipsoid.py:5-61        gm: a,b,c (1/semi-axis^2) */
    TRC_GM_ELLIPSOID_CUT = 27,     /* EllipsoidGM                ellipsoid.py:63-118      gm: a,b,c,xlo,xhi,ylo,yhi,zlo,zhi */
    TRC_GM_SPHERE_CUT = 28,        /* CutSphereGM                sphere_surface.py:168-204  gm: r, bound type (1 BoundaryPlane,
                                      2 BoundarySphere, 3 BoundaryCylinder; boundary_shape.py:89-162), bound rotation[9] and
                                      location[3] in the surface's frame, bound radius */
    TRC_GM_POLYGON = 29,           /* FlatSimplePolygonGM / PerforatedPolygonGM  polygon.py:8-53, :173-198  gm: n vertices, n holes,
                                      xmin,xmax,ymin,ymax ; extra: xs[n], ys[n] (clockwise, not closed), then n holes * (cx,cy,r) */
    TRC_GM_KIND_COUNT = 30
} trc_gm_kind;

/* ---- optics callables (reference: tracer/optics_callables.py) -------------- */
typedef enum trc_optics_kind {
    TRC_OPT_TRANSPARENT = 0,            /* Transparent             :93-113   */
    TRC_OPT_REFLECTIVE = 1,             /* Reflective              :116-140   opt: absorptivity ; Reflective_IAM :283-300 adds a_r (0: none), c */
    TRC_OPT_ONE_SIDED_REFLECTIVE = 2,   /* OneSidedReflective      :195-212   opt: absorptivity */
    TRC_OPT_REAL_REFLECTIVE = 3,        /* RealReflective          :214-269   opt: absorptivity,sigma,bi_var ; RealReflective_IAM :320-329 adds a_r, c */
    TRC_OPT_ONE_SIDED_REAL_REFLECTIVE = 4, /* OneSidedRealReflective :492-504 opt: absorptivity,sigma,bi_var */
    TRC_OPT_LAMBERTIAN = 5,             /* Lambertian              :143-176   opt: absorptivity,ang_range ; LambertianAbsorbant
                                           :891-906 adds attenuation_coefficient (0: none), scaling ; Lambertian_IAM :302-318 then a_r (0: none), c */
    TRC_OPT_LAMBERTIAN_SPECULAR = 6,    /* LambertianSpecular      :553-585   opt: absorptivity,specularity */
    TRC_OPT_REFRACTIVE_HOMOGENOUS = 7,  /* RefractiveHomogenous    :1186-1296 opt: n1,n2,single_ray,sigma(<0:none) ;
                                           RefractiveTransmissiveHomogenous :1326-1348 adds a_c in n1, a_c in n2, scaling, 1 */
    TRC_OPT_REFLECTIVE_SPECTRAL = 8,    /* Reflective_spectral     :178-193   extra: lambda[n] | absorptance[n] */
    TRC_OPT_LAMBERTIAN_DIRECTIONAL = 9, /* Lambertian_directional_axisymmetric_piecewise :331-361
                                           extra: theta[n] | absorptance[n] ; opt[0] = 1: specular with probability opt[1]
                                           (LambertianSpecular_directional_... :427-455); opt[0] = 2: with probability
                                           extra[2n..3n) on the same angles (Lambertian_piecewise_Specular_... :457-487) */
    TRC_OPT_LAMBERTIAN_DIRECTIONAL_SPECTRAL = 10, /* ..._piecewise_spectral :363-391
                                           extra: n_theta, n_lambda, theta[], lambda[], absorptance[n_theta][n_lambda] */
    TRC_OPT_FRESNEL_CONDUCTOR = 11,     /* FresnelConductorHomogenous :1523-1558  opt: n1 ; extra: lambda[n] | n[n] | k[n] */
    TRC_OPT_SEMI_LAMBERTIAN = 12,       /* SemiLambertian          :506-531   opt: absorptivity,angular_range -- as the class
                                           describes itself: mirror for incidence angles above angular_range, Lambertian
                                           below (its __call__ indexes the direction array by rows, :525, and cannot run) */
    TRC_OPT_REFRACTIVE_SCATTERING = 13, /* RefractiveScatteringHomogenous :1350-1376 on Scattering :946-1036: RefractiveHomogenous
                                           (single_ray) in media that scatter; opt as REFRACTIVE_HOMOGENOUS;
                                           extra: s_c1, s_c2 (scattering coefficients, 1/m), g1, g2 (Henyey-Greenstein) */
    TRC_OPT_REFRACTIVE_MATERIAL = 14,   /* Refractive :726-858 between two media of wavelength-dependent complex index m = n + ik
                                           (ray_trace_utils/optical_constants.py materials), and RefractiveAbsorbant :908-944
                                           (Absorbant.attenuate :874-889 with k = Im m, optics.py:205-212).  Rays carry a complex index
                                           (trc_rays.ref_index_im), a wavelength, and the materials' indices at it (trc_rays.mat).
                                           opt: single_ray, sigma(<0:none), attenuate(0|1), scaling, k0, k1 (rows of trc_rays.mat of
                                           material_1, material_2).  Ordered engine, per-surface protocol and (one ray per hit) the
                                           streaming form of the fast engine. */
    TRC_OPT_LAMBERTIAN_POLYCHROMATIC = 15, /* Lambertian_directional_axisymmetric_piecewise_Polychromatic :393-425: every ray carries a
                                           spectrum (trc_rays.spectra over trc_rays.spec_wl); each sample is scaled by 1 - absorptance(theta_in,
                                           lambda_w), the ray energy is the trapezoid integral of the result.  extra as
                                           LAMBERTIAN_DIRECTIONAL_SPECTRAL.  Ordered engine, protocol, streaming form of the fast engine. */
    TRC_OPT_PERIODIC_BOUNDARY = 16,     /* PeriodicBoundary :690-723: the ray stops on the surface (a stub of energy 0 keeps the tree
                                           connected: block 0) and goes on, unchanged, from the hit point moved by `period` along
                                           the oriented normal (block 1).  opt: period.  The fast engines follow the moved ray;
                                           the ordered engine records both, the stub among the culled rays. */
    TRC_OPT_KIND_COUNT = 17
} trc_optics_kind;

/* surface flags */
#define TRC_SURF_CAPTURE_HITS 0x1  /* append every hit to the scene hit buffer (Location/Direction accountants) */
#define TRC_SURF_CAPTURE_LEAN 0x2  /* with CAPTURE_HITS: only the absorbed energy and the hit point of this surface's hits are wanted
                                      (Absorption + Location accountants, the "Receiver" classes): incident energy and direction are
                                      not written -- 36 instead of 68 bytes per hit -- and read back as the absorbed energy and 0 */

/*
 * One Surface of the flattened Assembly (reference: tracer/surface.py:6-112 +
 * its GeometryManager + its optics callable).  frame is Surface._temp_frame
 * (has_frame.py:70-75) rows 0..2, row-major: frame[4*r+k], k=0..2 rotation,
 * k=3 translation.
 */
typedef struct trc_surface_desc {
    int32_t gm_kind;
    int32_t optics_kind;
    int32_t flags;
    int32_t extra_off;   /* first index into the scene's `extra` array, or -1 */
    int32_t extra_len;   /* number of doubles */
    int32_t reserved;
    double frame[12];
    double gm[16];
    double opt[8];
} trc_surface_desc;

/*
 * A ray bundle as structure-of-arrays (reference: RayBundle, ray_bundle.py:6-195).
 * Required: x..e.  Optional (NULL when absent): parent, ref_index, wavelength, rid, ref_index_im, spec_wl + spectra, mat.
 * Rows of the 2-D columns are `n` apart, n being this struct's n when it is handed over (an output's capacity).
 * `rid` is the 64-bit random-stream id of a ray (RNG); when NULL
 * it is ray_offset + index.  on_device != 0 means the pointers are device
 * pointers valid in this process (e.g. torch tensors on the context's GPU).
 */
typedef struct trc_rays {
    int64_t n;
    int32_t on_device;
    int32_t n_spec;             /* polychromatic bundles (ray_bundle.py `spectra`, `wavelengths` as 2-D columns): samples per ray, 0 = none */
    double *x, *y, *z;
    double *dx, *dy, *dz;
    double *e;
    int64_t *parent;
    double *ref_index;
    double *wavelength;
    uint64_t *rid;
    double *ref_index_im;       /* imaginary part of a complex refractive index (attenuating media); NULL = 0 */
    double *spec_wl;            /* n_spec x n: wavelength of sample w of ray i at spec_wl[w * n + i] (n = the bundle's ray count) */
    double *spectra;            /* n_spec x n: spectral power, same layout; its trapezoid integral over spec_wl is the ray energy */
    int64_t n_mat;              /* TRC_OPT_REFRACTIVE_MATERIAL: number of materials, and                                          */
    double *mat;                /* 2 n_mat x n: Re m_k(lambda_i) at mat[2k * n + i], Im at mat[(2k + 1) * n + i] -- material k's own
                                   m() evaluated by the caller at the wavelength of ray i (children inherit the wavelength)     */
} trc_rays;

/* ---- sources (reference: tracer/sources.py) -------------------------------- */
typedef enum trc_source_kind {
    TRC_SRC_PILLBOX_DISK = 0,  /* disk_bundle    sources.py:175-239  p: radius,radius_in,span0,span1,ang_range,
                                  has_x_cut,x_cut (positions redrawn until local x < x_cut) */
    TRC_SRC_PILLBOX_RECT = 1,  /* rect_bundle    sources.py:241-264  p: x,y,ang_range,swap_xy */
    TRC_SRC_BUIE_DISK = 2,     /* buie_sunshape  sources.py:412-464  p: radius ; tables */
    TRC_SRC_BUIE_RECT = 3,     /* rect_buie_sunshape sources.py:466-515 p: width,height ; tables */
    TRC_SRC_PILLBOX_TRIANGLE = 4, /* triangular_bundle sources.py:544-597  center = A, rot_pos columns 0,1 = AB, AC ; p: ang_range */
    TRC_SRC_VF_CYLINDER = 5,   /* vf_cylinder_bundle sources.py:716-769  Lambertian emitter on a cylinder wall
                                  p: rc,lc,span0,span1,ang_range,sign(+1 rays_in / -1) */
    TRC_SRC_VF_FRUSTUM = 6     /* vf_frustum_bundle sources.py:644-714  Lambertian emitter on a frustum wall
                                  p: r0,r1,depth,span0,span1,ang_range,sign */
} trc_source_kind;

#define TRC_BUIE_NELEM 210  /* sources.py:338 */

/*
 * Source descriptor: everything that does not depend on the random draws is
 * evaluated once on the host (rotation_to_z frames spatial_geometry.py:24-48,
 * per-ray energy, the Buie CDF table sources.py:333-361) and passed here.
 * buie[] layout: theta[211] | g[211] (=phi*cos*sin) | cdf[211] |
 *   I_dni, gamma, kappa, theta_dni, theta_tot, csr_positive.
 */
typedef struct trc_source_desc {
    int32_t kind;
    int32_t reserved;
    double center[3];
    double rot_pos[9];  /* row-major local->global for start points */
    double rot_dir[9];  /* row-major local->global for directions   */
    double p[8];
    double energy;      /* energy carried by each ray */
    double buie[3 * (TRC_BUIE_NELEM + 1) + 6];
} trc_source_desc;

/* ---- Kd-tree (reference: tracer/accel_tree.py) ------------------------------ */
/*
 * Flattened tree as built on the host with the reference's SAH rules
 * (accel_tree.py:42-204).  flag: 0/1/2 = split axis, 3 = leaf.  Interior:
 * split + child (children are child, child+1).  Leaf: leaf_off/leaf_cnt into
 * leaf_surfs.  always_relevant: surfaces of objects without boundaries
 * (accel_tree.py:59-73).  bounds: root box min xyz, max xyz.
 */
typedef struct trc_kdtree_desc {
    int32_t n_nodes;
    int32_t n_leaf_surfs;
    int32_t n_always;
    int32_t reserved;
    const int32_t *flag;
    const double *split;
    const int32_t *child;
    const int32_t *leaf_off;
    const int32_t *leaf_cnt;
    const int32_t *leaf_surfs;
    const int32_t *always_relevant;
    double bounds[6];
} trc_kdtree_desc;

typedef struct trc_ctx trc_ctx;
typedef struct trc_scene trc_scene;
typedef struct trc_result trc_result;

/* trace flags */
#define TRC_TRACE_ACCEL 0x1        /* accelerated candidate search (ray_tracer(accel=...)): the Kd-tree set on the scene
                                      (ordered engine, megakernel) or the library's own uniform grid (streaming form,
                                      no tree needed); results equal brute force either way */
#define TRC_TRACE_KEEP_LAST 0x2    /* fast engine: keep rays still alive after `reps` bounces */
#define TRC_TRACE_STREAM 0x4       /* fast engine: always run the phases as separate kernels connected by HBM queues
                                      (the default for calls of 1048576 rays and more); same results */
#define TRC_TRACE_MEGAKERNEL 0x8   /* fast engine: always run the persistent single-launch kernel (the default for
                                      smaller calls); same results */

typedef struct trc_trace_stats {
    int64_t segments;     /* sum over bounces of live rays (SURVEY 8(d) unit of work) */
    int64_t hits;         /* segments that hit a surface */
    int64_t rays_left;    /* rays still alive after the last bounce */
    int64_t hits_dropped; /* hits not captured because the hit buffer was full */
    double energy_left;
    double kernel_ms;     /* device time of the trace kernels (HIP events on the launch stream) */
    int32_t bounces;      /* iterations executed */
    int32_t launches;     /* kernel launches issued */
} trc_trace_stats;

const char *trc_last_error(void);
int trc_abi_version(void);

/* context: one per GPU ------------------------------------------------------ */
int trc_ctx_create(int device_id, trc_ctx **out);
int trc_ctx_destroy(trc_ctx *ctx);
int trc_ctx_synchronize(trc_ctx *ctx);
int trc_ctx_device_name(trc_ctx *ctx, char *buf, int buflen);

/* scene = Assembly.get_surfaces() flattened (assembly.py:60-77) --------------- */
int trc_scene_create(trc_ctx *ctx, int32_t n_surf, const trc_surface_desc *surfs,
                     int32_t n_extra, const double *extra, trc_scene **out);
int trc_scene_destroy(trc_scene *scene);
/* re-aim: Assembly.transform_children() (assembly.py:135-146) */
int trc_scene_update_frames(trc_scene *scene, int32_t n_surf, const double *frames12);
/* KdTree(...) result (accel_tree.py:20-40) */
int trc_scene_set_kdtree(trc_scene *scene, const trc_kdtree_desc *kd);
/* energy-weighted 2-D histogram of absorbed energy in the local x,y of one surface: the caller-side
   numpy.histogram2d of examples/Sandia_NSTTF_field example.py:175-227 and RectPlateGM.get_fluxmap
   (flat_surface.py:237-251), accumulated on the device.  u_edges[nu+1], v_edges[nv+1] are the numpy
   bin edges (last bin closed on the right); proj12 = rows 0..2 of the global->local matrix, i.e.
   round(inv(frame), 9) as in Surface.global_to_local (surface.py:114-126).  Resets the tallies. */
int trc_scene_set_fluxmap(trc_scene *scene, int32_t surf, int32_t nu, int32_t nv,
                          const double *u_edges, const double *v_edges, const double *proj12);
/* capacity (in hits) of the hit buffer that backs the Location/Direction/
   Absorption accountants (optics_callables.py:1597-1771) in the fast engine */
int trc_scene_set_hit_capacity(trc_scene *scene, int64_t capacity);
/* empty the hit buffer (the accountants' per-trace lists live on the Python side) */
int trc_scene_clear_hits(trc_scene *scene);
/* A hit buffer of at least `capacity` hits that keeps what it holds (set_hit_capacity starts an empty one): the accountants of
   the reference accumulate over calls until they are reset (optics_callables.py:1577-1643), so a script that traces again
   before it has read the hits of its last call leaves them on the device and reads them all at once later. */
int trc_scene_reserve_hits(trc_scene *scene, int64_t capacity);
/* entries of the hit buffer reserved so far -- written hits and the unused parts of the chunks the streaming engine keeps
   open -- and the capacity last asked for.  Either pointer may be NULL. */
int trc_scene_hits_reserved(trc_scene *scene, int64_t *reserved, int64_t *capacity);
/* Page-locked host memory for the large results of this library (the hit lists behind get_all_hits(), the levels of
   engine.tree, trace_tree.py:6-55): device-to-host copies into it run at the rate of the link.  Freed blocks are kept for
   the next request of their size. */
int trc_host_alloc(int64_t bytes, void **out);
int trc_host_free(void *p);
/* Assembly.reset_all_optics() (assembly.py:148-151) */
int trc_scene_reset_tallies(trc_scene *scene);
/* per-surface totals: absorbed = sum(E_in - sum E_out) (AbsorptionAccountant :1638-1643),
   received = sum E_in (ReceptionAccountant :1699-1701), hits = count. Any pointer may be NULL. */
int trc_scene_get_tallies(trc_scene *scene, double *absorbed, double *received, int64_t *hits);
int trc_scene_get_fluxmap(trc_scene *scene, int32_t surf, double *out /* nu*nv, row-major u */);
/* captured hits: in device arrival order when one surface captures; when several do, surface by surface (ascending index),
   arrival order inside a surface. Query n first with all arrays NULL -- or hand over arrays with room for every entry
   reserved so far (trc_scene_hits_reserved: an upper bound of n) and read n afterwards. */
int trc_scene_get_hits(trc_scene *scene, int64_t *n, int32_t *surf, double *e_abs, double *e_in,
                       double *px, double *py, double *pz, double *dx, double *dy, double *dz);
/* The same with the spectra of polychromatic hits (trc_trace_fast on a bundle with spectra, k_s_shade_x): n_x = 3 W more columns per
   hit, x[k * n + i] for hit i -- k in [0, W): sample wavelengths, [W, 2W): the spectrum that arrived, [2W, 3W): the spectrum that
   left (optics_callables.py:1825-1848 takes their difference).  trc_scene_hit_spectral_columns: the n_x the buffer holds (0: none). */
int trc_scene_get_hits_x(trc_scene *scene, int64_t *n, int32_t *surf, double *e_abs, double *e_in, double *px, double *py,
                         double *pz, double *dx, double *dy, double *dz, int32_t n_x, double *x);
int trc_scene_hit_spectral_columns(trc_scene *scene, int32_t *n_x);
/* View-factor allocation (emissive_losses/view_factors_3D.py:239-356 and :598-674, `alloc_VF`): the absorbed energy of
   the captured hits collected per element on the device instead of fetching every hit and looping over the elements on
   the host.  Element j takes the hits of the surfaces surf_lo[j]..surf_hi[j] whose global azimuth atan2(y,x) (brought
   to [0, 2 pi), :304-305), height z and radius sqrt(x^2+y^2) lie in ranges6[6j..6j+5] = {ang0, ang1, h0, h1, r0, r1}.
   mode[j] bits: TRC_BIN_ANGLE / TRC_BIN_HEIGHT / TRC_BIN_RADIUS select the tests (closed ranges, so that a hit on a
   shared edge counts in both elements, as in the reference); TRC_BIN_ROUND9 rounds height and radius to 9 decimals
   first (numpy.around, :308 and :633-634); TRC_BIN_RADIUS_HALF_OPEN makes the radius test r0 <= r < r1 (cone elements,
   :669).  out[n_bins] is overwritten.  The hit buffer is left as it is. */
#define TRC_BIN_ANGLE 0x1
#define TRC_BIN_HEIGHT 0x2
#define TRC_BIN_RADIUS 0x4
#define TRC_BIN_ROUND9 0x8
#define TRC_BIN_RADIUS_HALF_OPEN 0x10
int trc_scene_bin_hits(trc_scene *scene, int32_t n_bins, const int32_t *surf_lo, const int32_t *surf_hi,
                       const double *ranges6, const int32_t *mode, double *out);
/* Surface-to-surface energy transfer of the fast engine: T[from][to] = energy carried by the segments that leave
   surface `from` (row n_surf = the source) and land on surface `to`, (n_surf+1) x n_surf, row-major.  It replaces
   the blocking / shading post-process of examples/Sandia_NSTTF_field example.py:229-290, which recovers the parent
   heliostat of every blocked ray by comparing hit coordinates for equality, O(hits^2) on the host:
   incoming[h] = T[source][h] (:283), blocking[h] = sum over heliostats h' of T[h][h'] (:277), and T[h][receiver] is
   the contribution of heliostat h to the receiver.  Scenes of up to 1024 surfaces.  Enabling or disabling resets
   the tallies; the matrix travels at the end of the packed tally buffer (one all-reduce covers it). */
int trc_scene_enable_transfer(trc_scene *scene, int32_t on);
int trc_scene_get_transfer(trc_scene *scene, double *out /* (n_surf+1)*n_surf */);
/* the packed float64 tally buffer [absorbed S | received S | hits S | segments,hits | flux maps | transfer]
   for the single end-of-run reduce across GPUs (reference merge: tracer_engine_mp.py:44-119).
   export/import copy to/from a caller buffer (host, or device when on_device != 0) so the
   caller can run ncclAllReduce / torch.distributed.all_reduce on it. */
int trc_scene_tally_size(trc_scene *scene, int64_t *n_doubles);
int trc_scene_export_tallies(trc_scene *scene, double *dst, int32_t on_device);
int trc_scene_import_tallies(trc_scene *scene, const double *src, int32_t on_device);

/*
 * KdTree.traversal(bundle) (accel_tree.py:213-312 on intersect_bounds :314-330): which surfaces each ray has to be tested
 * against.  relevancy[s * n + r] = 1 when ray r crosses a leaf that holds surface s -- every leaf on its way through the
 * root box, not only the first -- or when s is always relevant (no bounds).  Host arrays; relevancy holds n_surf * n bytes.
 * *any_inter (may be NULL): the first value the reference returns.  Host bundles.
 */
int trc_kdtree_traversal(trc_ctx *ctx, const trc_kdtree_desc *kd, int32_t n_surf, const trc_rays *rays, int64_t n,
                         uint8_t *relevancy, int32_t *any_inter);

/*
 * TracerEngine.ray_tracer(bundle, reps, min_energy, tree=False, accel) (tracer_engine.py:124-295):
 * the persistent-wavefront engine.  Exactly one of `in` / `src` is non-NULL: `in` traces a given
 * bundle; `src` fuses source generation (sources.py) into the kernel, ray i of the call has
 * stream id ray_offset+i.  Tallies/flux maps/hit buffer accumulate on the scene.
 * If TRC_TRACE_KEEP_LAST, rays still alive after `reps` bounces are written to `last`
 * (capacity last->n on entry, count on exit).
 * A given bundle may carry what Refractive / RefractiveAbsorbant and the polychromatic wall read (ref_index_im, mat, spec_wl +
 * spectra): such calls, and calls on scenes with those optics, run the streaming form (64 rays or more; TRC_ERR_UNSUPPORTED with
 * TRC_TRACE_MEGAKERNEL or fewer rays: use trc_trace_ordered).  Captured hits keep their spectra (trc_scene_get_hits_x); `last` does not.
 */
int trc_trace_fast(trc_scene *scene, const trc_rays *in, const trc_source_desc *src, int64_t n,
                   int32_t reps, double min_energy, uint64_t seed, uint64_t ray_offset,
                   int32_t flags, trc_rays *last, trc_trace_stats *stats);

/*
 * TracerEngine.ray_tracer(..., tree=True): the ordered engine.  Reproduces the reference's
 * bundle ordering (surface-major, culled rays after live ones: tracer_engine.py:218-274) and
 * parent indices, one RayTree level (trace_tree.py:6-55) per bounce, kept on the device until
 * fetched.  Level 0 is the source bundle.
 */
int trc_trace_ordered(trc_scene *scene, const trc_rays *in, const trc_source_desc *src, int64_t n,
                      int32_t reps, double min_energy, uint64_t seed, uint64_t ray_offset,
                      int32_t flags, trc_result **out, trc_trace_stats *stats);
int trc_result_num_levels(trc_result *res, int32_t *n_levels);
/* n_total rays recorded at this level, the first n_live of which continued to the next bounce */
int trc_result_level_size(trc_result *res, int32_t level, int64_t *n_total, int64_t *n_live);
/* copy a level to host arrays (capacity out->n >= n_total); surf[i] = surface that produced ray i (-1 at level 0) */
/* surf[] of a level: the surface each ray of the level comes from.  TRC_LEVEL_VOLUME set: the ray never reached that surface --
   it was scattered in the medium in front of it (RefractiveScatteringHomogenous, optics_callables.py:946-1036) and is filed
   in the surface's scattered block; accountants of the surface do not see it. */
#define TRC_LEVEL_VOLUME 0x40000000
int trc_result_level_get(trc_result *res, int32_t level, trc_rays *out, int32_t *surf);
int trc_result_destroy(trc_result *res);

/* sources.*_bundle(...) materialised as a bundle (sources.py:175-515) */
int trc_source_generate(trc_ctx *ctx, const trc_source_desc *src, int64_t n, uint64_t seed,
                        uint64_t ray_offset, trc_rays *out);

/*
 * Start points of the first n rays of a disc / rectangle source (sources.py:175-515) in the source's own plane coordinates
 * (the two columns of rot_pos), evaluated in float32 exactly as the fast engine's culling kernel evaluates them before it
 * decides from the footprint map (csrc/trc_footprint.h) whether a ray can reach any surface.  *eps receives the distance the
 * map allows between this and the float64 start point of trc_source_generate.  A self-check of the engine: tests compare
 * the two on the device.  TRC_ERR_UNSUPPORTED for sources the map does not apply to.
 */
int trc_source_start32(trc_ctx *ctx, const trc_source_desc *src, int64_t n, uint64_t seed,
                       uint64_t ray_offset, float *lx, float *ly, double *eps);

/*
 * Per-surface trace protocol (user-doc/trace_protocol.rst:1-24), for callers that drive
 * surfaces one at a time like the reference engine does:
 *   GeometryManager.find_intersections(frame, bundle) -> t (+inf = miss)   geometry_manager.py:8-26
 *   GeometryManager.get_normals() / get_intersection_points_global()       quadric.py:159-172
 *   optics(geometry, rays, selector)                                       surface.py:84-94
 */
int trc_gm_find_intersections(trc_ctx *ctx, const trc_surface_desc *surf, int32_t n_extra,
                              const double *extra, const trc_rays *rays, double *t_out,
                              double *hx, double *hy, double *hz);
int trc_gm_get_normals(trc_ctx *ctx, const trc_surface_desc *surf, int64_t n,
                       const double *hx, const double *hy, const double *hz,
                       const double *dx, const double *dy, const double *dz,
                       double *nx, double *ny, double *nz);
/*
 * optics callable on n selected hits. in: incident rays (direction, energy, ref_index,
 * wavelength, rid; origins x,y,z when the optics attenuates along the path: Absorbant.attenuate
 * :874-889) + hit points + oriented normals.  out: capacity 2n rays; out->n is set to the
 * number produced; out->parent[k] indexes the n inputs; block order as the reference's
 * (reflected block, then refracted block: optics_callables.py:1284-1294).
 */
int trc_optics_apply(trc_ctx *ctx, const trc_surface_desc *surf, int32_t n_extra, const double *extra,
                     const trc_rays *in, const double *hx, const double *hy, const double *hz,
                     const double *nx, const double *ny, const double *nz,
                     uint64_t seed, int32_t bounce, trc_rays *out);

/*
 * optics.fresnel_to_attenuating (tracer/optics.py:63-81): reflectances R_p, R_s and refraction angle at the
 * interface between a dielectric of index n1 and an absorbing medium of complex index m_re + i m_im, for n
 * incidence angles theta1 (radians).  Host arrays of length n.
 */
int trc_optics_fresnel_attenuating(trc_ctx *ctx, int64_t n, double n1, const double *m_re, const double *m_im,
                                   const double *theta1, double *r_p, double *r_s, double *theta2);

#ifdef __cplusplus
}
#endif
#endif /* TRACER_AMD_H */
