/*
 * pandora_oracle.c -- TEST INFRASTRUCTURE ONLY (see pandora_oracle.h for the pinning status).
 *
 * Scalar restatement of the reference's traversal path. Every function cites the reference
 * file:line it follows (paths relative to /root/reference/projects/).
 * Build: gcc -O2 -ffp-contract=off (no -ffast-math): float results must not depend on FMA fusion.
 */
#include "pandora_oracle.h"
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* x86 minps/maxps: second operand wins when either is NaN (simd8_avx2.ipp:439-451) */
static inline float min86(float a, float b) { return a < b ? a : b; }
static inline float max86(float a, float b) { return a > b ? a : b; }
/* std::min / std::max (used by the scalar parts of sparse_voxel_dag.cpp:325-348,398) */
static inline float stdmin(float a, float b) { return b < a ? b : a; }
static inline float stdmax(float a, float b) { return a < b ? b : a; }

static inline uint32_t f2u(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float u2f(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* ---------------------------------------------------------------- simd primitives (S4) */

/* simd8_avx2.ipp:1-17: byte j of the LUT entry = index of the j-th set bit, remaining bytes 0 */
void orc_compress_perm8(unsigned mask, uint32_t out[8])
{
    unsigned k = 0;
    for (unsigned j = 0; j < 8; j++)
        out[j] = 0;
    for (unsigned bit = 0; bit < 8; bit++)
        if (mask & (1u << bit))
            out[k++] = bit;
}

/* intrinsics.h:22-34 */
int orc_bitscan4(unsigned mask)
{
    if (mask & 1u)
        return 0;
    if (mask & 2u)
        return 1;
    if (mask & 4u)
        return 2;
    return 3;
}

/* simd4_avx2.ipp:412-430: two shuffle+minps rounds, then cmpeq against the lane-wise result */
unsigned orc_hmin_index4(const float v[4])
{
    float m2[4], m4[4];
    const int s1[4] = { 2, 3, 0, 1 }; /* _MM_SHUFFLE(1,0,3,2) */
    const int s2[4] = { 1, 0, 3, 2 }; /* _MM_SHUFFLE(2,3,0,1) */
    for (int i = 0; i < 4; i++)
        m2[i] = min86(v[i], v[s1[i]]);
    for (int i = 0; i < 4; i++)
        m4[i] = min86(m2[i], m2[s2[i]]);
    unsigned eq = 0;
    for (int i = 0; i < 4; i++)
        if (v[i] == m4[i])
            eq |= 1u << i;
    return (unsigned)orc_bitscan4(eq);
}

/* intrinsics.h:93-126 (non-MSVC branch): index of the highest set bit */
int orc_bsr32(uint32_t mask)
{
    int r = 0;
    while (mask >>= 1)
        r++;
    return r;
}

/* wive_bvh8_impl.h:384-388: zero counts as negative */
uint32_t orc_sign_shift(float dx, float dy, float dz)
{
    return ((dx > 0 ? 1u : 0u) | (dy > 0 ? 2u : 0u) | (dz > 0 ? 4u : 0u)) * 3u;
}

/* wive_bvh8_build8_impl.h:172-203. std::sort on 8 elements is libstdc++'s insertion sort, i.e.
 * stable; ties (and the FLT_MAX padding of absent children) therefore keep index order. */
void orc_bvh8_perm(const float lo[][3], const float hi[][3], unsigned n_children, uint32_t perm[8])
{
    for (int i = 0; i < 8; i++)
        perm[i] = 0;
    for (int x = -1; x <= 1; x += 2)
        for (int y = -1; y <= 1; y += 2)
            for (int z = -1; z <= 1; z += 2) {
                float dist[8];
                uint32_t order[8];
                for (unsigned i = 0; i < 8; i++) {
                    order[i] = i;
                    if (i >= n_children) {
                        dist[i] = FLT_MAX;
                        continue;
                    }
                    float ex = x == -1 ? lo[i][0] : hi[i][0];
                    float ey = y == -1 ? lo[i][1] : hi[i][1];
                    float ez = z == -1 ? lo[i][2] : hi[i][2];
                    /* glm::dot(vec3,vec3): (x*x' + y*y') + z*z' */
                    dist[i] = (ex * (float)x + ey * (float)y) + ez * (float)z;
                }
                for (int i = 1; i < 8; i++) { /* stable insertion sort, descending */
                    uint32_t v = order[i];
                    int j = i - 1;
                    while (j >= 0 && dist[v] > dist[order[j]]) {
                        order[j + 1] = order[j];
                        j--;
                    }
                    order[j + 1] = v;
                }
                uint32_t shift = ((x > 0 ? 1u : 0u) | (y > 0 ? 2u : 0u) | (z > 0 ? 4u : 0u)) * 3u;
                for (int i = 0; i < 8; i++)
                    perm[i] |= order[i] << shift;
            }
}

/* ---------------------------------------------------------------- triangle tests (L1, L2) */

static inline float dot3(const float a[3], const float b[3])
{
    /* glm compute_dot<vec3>: tmp = a*b; tmp.x + tmp.y + tmp.z */
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
static inline void cross3(const float x[3], const float y[3], float o[3])
{
    /* glm compute_cross */
    o[0] = x[1] * y[2] - y[1] * x[2];
    o[1] = x[2] * y[0] - y[2] * x[0];
    o[2] = x[0] * y[1] - y[0] * x[1];
}

/* shapes/triangle_intersect.cpp:425-462 */
static int tri_mt(const float o[3], const float d[3], float tnear, float* tfar,
    const float p0[3], const float p1[3], const float p2[3], float* u_out, float* v_out)
{
    const float EPS = 0.000001f;
    float e1[3], e2[3], h[3], s[3], q[3];
    for (int i = 0; i < 3; i++) {
        e1[i] = p1[i] - p0[i];
        e2[i] = p2[i] - p0[i];
    }
    cross3(d, e2, h);
    float a = dot3(e1, h);
    if (a > -EPS && a < EPS)
        return 0;
    float f = 1.0f / a;
    for (int i = 0; i < 3; i++)
        s[i] = o[i] - p0[i];
    float u = f * dot3(s, h);
    if (u < 0.0f || u > 1.0f)
        return 0;
    cross3(s, e1, q);
    float v = f * dot3(d, q);
    if (v < 0.0f || u + v > 1.0f)
        return 0;
    float t = f * dot3(e2, q);
    if (t < tnear || t > *tfar)
        return 0;
    *tfar = t;
    *u_out = u;
    *v_out = v;
    return 1;
}

/* shapes/triangle_intersect.cpp:340-423 (PBRT-v3 watertight test, compiled out in the reference) */
static int tri_watertight(const float o[3], const float d[3], float tnear, float* tfar,
    const float p0[3], const float p1[3], const float p2[3], float* u_out, float* v_out)
{
    float a[3][3];
    for (int i = 0; i < 3; i++) {
        a[0][i] = p0[i] - o[i];
        a[1][i] = p1[i] - o[i];
        a[2][i] = p2[i] - o[i];
    }
    /* maxDimension(abs(d)): utility/math.h -- (x > y) ? (x > z ? 0 : 2) : (y > z ? 1 : 2) */
    float ax = fabsf(d[0]), ay = fabsf(d[1]), az = fabsf(d[2]);
    int kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
    int kx = kz + 1;
    if (kx == 3)
        kx = 0;
    int ky = kx + 1;
    if (ky == 3)
        ky = 0;
    float dp[3] = { d[kx], d[ky], d[kz] };
    float t[3][3];
    for (int k = 0; k < 3; k++) {
        t[k][0] = a[k][kx];
        t[k][1] = a[k][ky];
        t[k][2] = a[k][kz];
    }
    float Sx = -dp[0] / dp[2];
    float Sy = -dp[1] / dp[2];
    float Sz = 1.0f / dp[2];
    for (int k = 0; k < 3; k++) {
        t[k][0] += Sx * t[k][2];
        t[k][1] += Sy * t[k][2];
    }
    float e0 = t[1][0] * t[2][1] - t[1][1] * t[2][0];
    float e1 = t[2][0] * t[0][1] - t[2][1] * t[0][0];
    float e2 = t[0][0] * t[1][1] - t[0][1] * t[1][0];
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double p2txp1ty = (double)t[2][0] * (double)t[1][1];
        double p2typ1tx = (double)t[2][1] * (double)t[1][0];
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)t[0][0] * (double)t[2][1];
        double p0typ2tx = (double)t[0][1] * (double)t[2][0];
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)t[1][0] * (double)t[0][1];
        double p1typ0tx = (double)t[1][1] * (double)t[0][0];
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f))
        return 0;
    float det = e0 + e1 + e2;
    if (det == 0.0f)
        return 0;
    t[0][2] *= Sz;
    t[1][2] *= Sz;
    t[2][2] *= Sz;
    float tScaled = e0 * t[0][2] + e1 * t[1][2] + e2 * t[2][2];
    if (det < 0.0f && (tScaled >= tnear * det || tScaled < *tfar * det))
        return 0;
    else if (det > 0.0f && (tScaled <= tnear * det || tScaled > *tfar * det))
        return 0;
    float invDet = 1.0f / det;
    *u_out = e0 * invDet;
    *v_out = e1 * invDet;
    *tfar = tScaled * invDet;
    return 1;
}

int orc_tri_intersect(int mode, const float o[3], const float d[3], float tnear, float* tfar,
    const float p0[3], const float p1[3], const float p2[3], float* u, float* v)
{
    return mode == ORC_TRI_WATERTIGHT ? tri_watertight(o, d, tnear, tfar, p0, p1, p2, u, v)
                                      : tri_mt(o, d, tnear, tfar, p0, p1, p2, u, v);
}

/* ---------------------------------------------------------------- BVH8 (B2-B7) */

#define H_FLAGS(h) (((h) >> 29) & 7u)
#define H_IS_INNER(h) (H_FLAGS(h) == 2u)
#define H_INDEX(h) ((h)&0x1FFFFFFFu)
#define H_LEAF_COUNT(h) ((((h) >> 29) & 3u) + 1u)

/* wive_bvh8_impl.h:209-238 */
unsigned orc_bvh8_inner(const orc_bvh8_node* n, const orc_ray* r, uint32_t out_children[8], float out_dist[8])
{
    const float ix = 1.0f / r->d[0], iy = 1.0f / r->d[1], iz = 1.0f / r->d[2]; /* :66-68, no zero guard */
    const uint32_t shift = orc_sign_shift(r->d[0], r->d[1], r->d[2]);
    float tmin[8], tmax[8];
    for (int c = 0; c < 8; c++) {
        float tx1 = (n->minX[c] - r->o[0]) * ix, tx2 = (n->maxX[c] - r->o[0]) * ix;
        float ty1 = (n->minY[c] - r->o[1]) * iy, ty2 = (n->maxY[c] - r->o[1]) * iy;
        float tz1 = (n->minZ[c] - r->o[2]) * iz, tz2 = (n->maxZ[c] - r->o[2]) * iz;
        float txMin = min86(tx1, tx2), tyMin = min86(ty1, ty2), tzMin = min86(tz1, tz2);
        float txMax = max86(tx1, tx2), tyMax = max86(ty1, ty2), tzMax = max86(tz1, tz2);
        tmin[c] = max86(r->tnear, max86(txMin, max86(tyMin, tzMin)));
        tmax[c] = min86(r->tfar, min86(txMax, min86(tyMax, tzMax)));
    }
    unsigned k = 0;
    for (int j = 0; j < 8; j++) { /* slot j takes child (perm[j] >> shift) & 7; kept slots keep their order */
        uint32_t c = (n->perm[j] >> shift) & 7u;
        if (tmin[c] <= tmax[c]) {
            out_children[k] = n->children[c];
            out_dist[k] = tmin[c];
            k++;
        }
    }
    return k;
}

static inline const float* tri_vertex(const orc_mesh* m, uint32_t prim, int k)
{
    return m->pos + 3u * (size_t)m->idx[3u * (size_t)prim + (unsigned)k];
}

/* Transform::transformToLocal, graphics_core/transform.cpp:40-45: origin = inverse * (o, 1), direction = inverse * (d, 0),
 * tnear / tfar unchanged. glm's mat4 * vec4 adds the four column products as (c0 x + c1 y) + (c2 z + c3 w) (glm is not in the
 * container: restated from its published source, type_mat4x4.inl). */
void orc_ray_to_local(const float m[12], const orc_ray* in, orc_ray* out)
{
    for (int i = 0; i < 3; i++) {
        const float* row = m + 4 * i;
        out->o[i] = (row[0] * in->o[0] + row[1] * in->o[1]) + (row[2] * in->o[2] + row[3] * 1.0f);
        out->d[i] = (row[0] * in->d[0] + row[1] * in->d[1]) + (row[2] * in->d[2] + row[3] * 0.0f);
    }
    out->tnear = in->tnear;
    out->tfar = in->tfar;
}

/* wive_bvh8_impl.h:272-280 + offline_bvh_cache.cpp:42-66. Primitive alternative: the triangle test. Instance alternative
 * (:54-65): the nested BVH's intersect. The reference hands the nested traversal the UNTRANSFORMED ray -- the offline path is
 * unfinished there; what its other acceleration structures do for the same scene graph
 * (batching_acceleration_structure.h:411-422, embree_acceleration_structure.h:143-153) is restated here: the ray goes to the
 * instance's local space with Transform::transformToLocal, t keeps its meaning. REF-UNPINNED. */
static int bvh8_leaf(const orc_bvh8* b, int tri_mode, uint32_t handle, orc_ray* r, orc_hit* h, orc_counters* c)
{
    const uint32_t* ids = b->leaf_idx + H_INDEX(handle);
    const uint32_t count = H_LEAF_COUNT(handle);
    int hit = 0;
    for (uint32_t i = 0; i < count; i++) {
        uint32_t obj = ids[i];
        if (b->inst && b->inst[obj].child) {
            orc_ray local;
            orc_hit nested;
            orc_ray_to_local(b->inst[obj].world_to_local, r, &local);
            if (orc_bvh8_intersect(b->inst[obj].child, tri_mode, &local, &nested, c)) {
                hit = 1;
                r->tfar = local.tfar;
                if (h) {
                    h->t = local.tfar;
                    h->obj = nested.obj;
                    h->u = nested.u;
                    h->v = nested.v;
                    h->inst = obj;
                }
            }
            continue;
        }
        uint32_t prim = b->obj_prim ? b->obj_prim[obj] : obj;
        float u, v;
        if (c)
            c->tri_tests++;
        if (orc_tri_intersect(tri_mode, r->o, r->d, r->tnear, &r->tfar,
                tri_vertex(&b->mesh, prim, 0), tri_vertex(&b->mesh, prim, 1), tri_vertex(&b->mesh, prim, 2), &u, &v)) {
            hit = 1;
            if (h) {
                h->t = r->tfar;
                h->obj = obj;
                h->u = u;
                h->v = v;
                h->inst = ORC_MISS;
            }
        }
    }
    return hit;
}

#define ORC_STACK 96 /* wive_bvh8_impl.h:74-75 */

/* wive_bvh8_impl.h:57-140. The reference compacts the whole stack after every leaf hit keeping
 * dist < tfar (:112-134); order is preserved, so a plain in-order filter is equivalent. */
int orc_bvh8_intersect(const orc_bvh8* b, int tri_mode, orc_ray* r, orc_hit* h, orc_counters* c)
{
    uint32_t sh[ORC_STACK + 8];
    float sd[ORC_STACK + 8];
    size_t sp = 0;
    int hit = 0;
    sh[sp] = b->root;
    sd[sp] = 0.0f;
    sp++;
    while (sp > 0) {
        sp--;
        uint32_t handle = sh[sp];
        if (H_IS_INNER(handle)) {
            uint32_t ch[8];
            float di[8];
            if (c)
                c->inner_visits++;
            unsigned n = orc_bvh8_inner(&b->nodes[H_INDEX(handle)], r, ch, di);
            if (sp + 8 > ORC_STACK)
                abort(); /* the reference would write past its 96-entry arrays */
            for (unsigned i = 0; i < n; i++) {
                sh[sp + i] = ch[i];
                sd[sp + i] = di[i];
            }
            sp += n;
        } else {
            /* release-build behaviour: an empty handle (0) is read as "leaf, 1 prim, index 0" (:104-110) */
            if (c)
                c->leaf_pops++;
            if (bvh8_leaf(b, tri_mode, handle, r, h, c)) {
                hit = 1;
                size_t out = 0;
                for (size_t i = 0; i < sp; i++)
                    if (sd[i] < r->tfar) {
                        sh[out] = sh[i];
                        sd[out] = sd[i];
                        out++;
                    }
                sp = out;
            }
        }
    }
    return hit;
}

/* wive_bvh8_impl.h:142-201: same sorted inner routine, first hit wins, ray.tfar = -inf */
int orc_bvh8_intersect_any(const orc_bvh8* b, int tri_mode, orc_ray* r, orc_counters* c)
{
    uint32_t sh[ORC_STACK + 8];
    size_t sp = 0;
    sh[sp++] = b->root;
    while (sp > 0) {
        sp--;
        uint32_t handle = sh[sp];
        if (H_IS_INNER(handle)) {
            uint32_t ch[8];
            float di[8];
            if (c)
                c->inner_visits++;
            unsigned n = orc_bvh8_inner(&b->nodes[H_INDEX(handle)], r, ch, di);
            if (sp + 8 > ORC_STACK)
                abort();
            for (unsigned i = 0; i < n; i++)
                sh[sp + i] = ch[i];
            sp += n;
        } else {
            if (c)
                c->leaf_pops++;
            /* intersectAnyLeaf :282-290 -> OfflineBVHLeaf::intersectAny == intersectPrimitive */
            const uint32_t* ids = b->leaf_idx + H_INDEX(handle);
            const uint32_t count = H_LEAF_COUNT(handle);
            for (uint32_t i = 0; i < count; i++) {
                if (b->inst && b->inst[ids[i]].child) { /* offline_bvh_cache.cpp:74-77, ray taken to local space (see bvh8_leaf) */
                    orc_ray local;
                    orc_ray_to_local(b->inst[ids[i]].world_to_local, r, &local);
                    if (orc_bvh8_intersect_any(b->inst[ids[i]].child, tri_mode, &local, c)) {
                        r->tfar = -INFINITY;
                        return 1;
                    }
                    continue;
                }
                uint32_t prim = b->obj_prim ? b->obj_prim[ids[i]] : ids[i];
                float u, v;
                if (c)
                    c->tri_tests++;
                if (orc_tri_intersect(tri_mode, r->o, r->d, r->tnear, &r->tfar,
                        tri_vertex(&b->mesh, prim, 0), tri_vertex(&b->mesh, prim, 1), tri_vertex(&b->mesh, prim, 2), &u, &v)) {
                    r->tfar = -INFINITY;
                    return 1;
                }
            }
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- SVDAG (S1-S3) */

#define CAST_STACK_DEPTH 23

static inline uint32_t dag_valid(uint32_t desc) { return (desc >> 8) & 0xFFu; }
static inline uint32_t dag_leaf(uint32_t desc) { return desc & 0xFFu; }

/* sparse_voxel_dag.cpp:300-309 */
static inline uint32_t dag_child(const uint32_t* data, uint32_t node, uint32_t idx)
{
    uint32_t desc = data[node];
    uint32_t inner = dag_valid(desc) & ~dag_leaf(desc);
    uint32_t before = inner & ((1u << idx) - 1u);
    return data[node + 1u + (uint32_t)__builtin_popcount(before)];
}

/* sparse_voxel_dag.cpp:350-567. L = 4 reproduces the (x,y,z,1) vectors including the 0x07FFFFFF
 * mask of :475; L = 3 drops the fourth lane (what the GPU kernel does). */
int orc_svdag_intersect(const orc_svdag* dag, const orc_ray* ray, int L)
{
    const float eps = 1.1920928955078125e-07f;
    float o[4], d[4], tCoef[4], tBias[4], pos[4], tMinVec[4];
    const float ro[3] = { ray->o[0], ray->o[1], ray->o[2] };
    const float rd[3] = { ray->d[0], ray->d[1], ray->d[2] };
    for (int i = 0; i < 3; i++) {
        o[i] = 1.0f + (dag->inv_extent * (ro[i] - dag->bounds_min[i])); /* :352 */
        d[i] = rd[i];
        if (fabsf(d[i]) < eps)
            d[i] = copysignf(eps, d[i]); /* :359-364 */
    }
    o[3] = 1.0f;
    d[3] = 1.0f;
    uint32_t octMask4 = 0;
    for (int i = 0; i < L; i++) {
        tCoef[i] = 1.0f / -fabsf(d[i]); /* :370 */
        tBias[i] = tCoef[i] * o[i];
        if (d[i] > 0.0f) { /* :387-389 */
            octMask4 |= 1u << i;
            tBias[i] = 3.0f * tCoef[i] - tBias[i];
        }
    }
    const uint32_t octBits = (7u ^ octMask4) & 7u;

    uint32_t parent = dag->root;
    int scale = CAST_STACK_DEPTH - 1;
    float a3[3], b3[3];
    for (int i = 0; i < 3; i++) {
        a3[i] = 2.0f * tCoef[i] - tBias[i];
        b3[i] = tCoef[i] - tBias[i];
    }
    const float tMin = stdmax(0.0f, stdmax(a3[0], stdmax(a3[1], a3[2]))); /* :398 */
    const float tMax = stdmin(b3[0], stdmin(b3[1], b3[2]));                /* :399 */
    if (tMin >= tMax)
        return 0;

    uint32_t idx = 0;
    for (int i = 0; i < L; i++) { /* :422-424 */
        tMinVec[i] = tMin;
        int m = (1.5f * tCoef[i] - tBias[i]) > tMin;
        idx |= (uint32_t)m << i;
        pos[i] = m ? 1.5f : 1.0f;
    }

    uint32_t stack[CAST_STACK_DEPTH + 1];
    int hit = 0;
    /* height of the DAG: all leaves of an SVO sit at the same depth, follow first children */
    uint32_t levels = 1;
    for (uint32_t at = dag->root; levels < CAST_STACK_DEPTH; levels++) {
        const uint32_t dsc = dag->data[at];
        if ((dag_valid(dsc) & ~dag_leaf(dsc) & 0xFFu) == 0)
            break;
        at = dag->data[at + 1];
    }
    const uint64_t trip_cap = (uint64_t)(16u * levels + 16u) << levels;
    uint64_t trips = 0;
    while (scale < CAST_STACK_DEPTH) {
        if (++trips > trip_cap)
            return 1; /* no progress (NaN operands): give up, do not cull */
        float tCorner[4];
        for (int i = 0; i < L; i++)
            tCorner[i] = pos[i] * tCoef[i] - tBias[i]; /* :434 */
        const uint32_t childIndex = 7u - ((idx & 7u) ^ octBits);
        const uint32_t desc = dag->data[parent];
        if (dag_valid(desc) & (1u << childIndex)) {
            /* PUSH :439-470 */
            const float half = ldexpf(1.0f, (scale - 1) - CAST_STACK_DEPTH); /* scaleExp2LUT[scale-1] */
            stack[scale] = parent;
            if (dag_valid(desc) & dag_leaf(desc) & (1u << childIndex)) {
                hit = 1;
                break;
            }
            parent = dag_child(dag->data, parent, childIndex);
            scale--;
            idx = 0;
            for (int i = 0; i < L; i++) {
                float tCenter = half * tCoef[i] + tCorner[i];
                int m = tCenter > tMinVec[i];
                idx |= (uint32_t)m << i;
                pos[i] = m ? pos[i] + half : pos[i];
            }
            idx &= 7u;
        } else {
            /* ADVANCE :473-497 */
            const float scaleExp2 = ldexpf(1.0f, scale - CAST_STACK_DEPTH);
            float tcMax[4];
            if (L == 4) {
                float w[4] = { tCorner[0], tCorner[1], tCorner[2], u2f(f2u(tCorner[3]) | 0x07FFFFFFu) };
                float m2[4];
                const int s1[4] = { 2, 3, 0, 1 }, s2[4] = { 1, 0, 3, 2 }; /* horizontalMinVec, simd4_avx2.ipp:432-445 */
                for (int i = 0; i < 4; i++)
                    m2[i] = min86(w[i], w[s1[i]]);
                for (int i = 0; i < 4; i++)
                    tcMax[i] = min86(m2[i], m2[s2[i]]);
            } else {
                float m = stdmin(tCorner[0], stdmin(tCorner[1], tCorner[2]));
                tcMax[0] = tcMax[1] = tcMax[2] = m;
            }
            uint32_t step4 = 0;
            for (int i = 0; i < L; i++) {
                int s = tCorner[i] <= tcMax[i];
                step4 |= (uint32_t)s << i;
            }
            const uint32_t stepBits = step4 & 7u;
            float newPos[4];
            for (int i = 0; i < L; i++)
                newPos[i] = (step4 >> i & 1u) ? pos[i] - scaleExp2 : pos[i];
            for (int i = 0; i < L; i++) {
                pos[i] = newPos[i];
                tMinVec[i] = tcMax[i];
            }
            idx ^= stepBits;
            if ((idx & stepBits) != 0) {
                /* POP :500-526 */
                uint32_t diff = 0;
                for (int i = 0; i < 3; i++)
                    if (stepBits >> i & 1u)
                        diff |= f2u(pos[i]) ^ f2u(pos[i] + scaleExp2);
                scale = orc_bsr32(diff);
                if (scale <= CAST_STACK_DEPTH) /* the reference reads its 24-entry array unguarded */
                    parent = stack[scale];
            }
            /* :529-538 round position, rebuild slot index */
            uint32_t shv[4];
            for (int i = 0; i < L; i++) {
                shv[i] = f2u(pos[i]) >> scale;
                pos[i] = u2f(shv[i] << scale);
            }
            idx = (shv[0] & 1u) | ((shv[1] & 1u) << 1) | ((shv[2] & 1u) << 2)
                | ((shv[0] & 2u) << 2) | ((shv[1] & 2u) << 3) | ((shv[2] & 2u) << 4);
        }
    }
    return hit;
}

/* ---------------------------------------------------------------- top level BVH4 (P1-P6) */

#define N4_VALID(n) ((n)->bits & 0xFu)
#define N4_LEAF(n) (((n)->bits >> 4) & 0xFu)
#define N4_DEPTH(n) (((n)->bits >> 8) & 0xFFu)

/* pauseable_bvh4.h:225-355, one episode: from (node, stack) until the next batching point that
 * accepts the ray (pause, returns 1) or the exit (returns 0). BatchingPoint::intersect :269-286
 * supplies the SVDAG early-out. Deviation: children outside validMask are never
 * visited; the reference would index m_leafs with them. */
int orc_bvh4_step(const orc_scene* s, int any_hit, int svdag_lanes, const orc_ray* r,
    uint32_t* node_io, uint64_t* stack_io, uint32_t* leaf_out, orc_counters* c)
{
    return orc_bvh4_step_ex(s, any_hit, svdag_lanes, r, node_io, stack_io, leaf_out, c, 1);
}

int orc_bvh4_step_ex(const orc_scene* s, int any_hit, int svdag_lanes, const orc_ray* r,
    uint32_t* node_io, uint64_t* stack_io, uint32_t* leaf_out, orc_counters* c, int may_park)
{
    const uint64_t step_cap = 8ull * ((uint64_t)s->n_nodes + s->n_leaves) + 64ull;
    uint64_t steps = 0;
    const float ix = 1.0f / r->d[0], iy = 1.0f / r->d[1], iz = 1.0f / r->d[2]; /* :241 */
    uint32_t nodeHandle = *node_io;
    uint64_t stack = *stack_io;
    for (;;) {
        if (++steps > step_cap) { /* non-terminating traversal (see the header): the ray leaves the top level */
            stack = 0xFFFFFFFFFFFFFFFFull;
            break;
        }
        const orc_bvh4_node* n = &s->nodes[nodeHandle];
        const int bitPos = 4 * (int)N4_DEPTH(n);
        const uint64_t interest = (stack >> bitPos) & 0xFull;
        if (interest != 0) {
            float tmin[4];
            uint32_t hitMask = 0;
            for (int i = 0; i < 4; i++) {
                float tx1 = (n->minX[i] - r->o[0]) * ix, tx2 = (n->maxX[i] - r->o[0]) * ix;
                float ty1 = (n->minY[i] - r->o[1]) * iy, ty2 = (n->maxY[i] - r->o[1]) * iy;
                float tz1 = (n->minZ[i] - r->o[2]) * iz, tz2 = (n->maxZ[i] - r->o[2]) * iz;
                float txMin = min86(tx1, tx2), tyMin = min86(ty1, ty2), tzMin = min86(tz1, tz2);
                float txMax = max86(tx1, tx2), tyMax = max86(ty1, ty2), tzMax = max86(tz1, tz2);
                tmin[i] = max86(r->tnear, max86(txMin, max86(tyMin, tzMin)));
                float tmax = min86(r->tfar, min86(txMax, min86(tyMax, tzMax)));
                if (tmin[i] <= tmax)
                    hitMask |= 1u << i;
            }
            uint32_t toVisit = hitMask & (uint32_t)interest & N4_VALID(n);
            if (toVisit != 0) {
                unsigned child;
                if (any_hit) {
                    child = (unsigned)orc_bitscan4(toVisit); /* :295 */
                } else {
                    float masked[4];
                    for (int i = 0; i < 4; i++)
                        masked[i] = (toVisit >> i & 1u) ? tmin[i] : FLT_MAX; /* :297-299 */
                    child = orc_hmin_index4(masked);
                }
                uint64_t tv = toVisit;
                tv ^= (1ull << child);
                stack = stack ^ (interest << bitPos);
                stack = stack | (tv << bitPos);
                const uint32_t valid = N4_VALID(n), leafm = N4_LEAF(n);
                if ((valid & ~leafm) >> child & 1u) {
                    nodeHandle = n->firstChild + (uint32_t)__builtin_popcount((valid & ~leafm) & ((1u << child) - 1u));
                } else {
                    uint32_t leaf = n->firstLeaf + (uint32_t)__builtin_popcount((valid & leafm) & ((1u << child) - 1u));
                    const orc_leaf* lf = &s->leaves[leaf];
                    int culled = 0;
                    if (lf->dag.enabled) {
                        if (c)
                            c->svdag_tests++;
                        if (!orc_svdag_intersect(&lf->dag, r, svdag_lanes)) {
                            if (c)
                                c->svdag_culled++;
                            culled = 1;
                        }
                    }
                    if (!culled) {
                        if (!may_park) {
                            stack = 0xFFFFFFFFFFFFFFFFull;
                            break;
                        }
                        *node_io = nodeHandle;
                        *stack_io = stack;
                        *leaf_out = leaf;
                        return 1;
                    }
                }
                continue;
            }
        }
        stack = stack | (0xFFFFFFFFFFFFFFFFull << bitPos);
        if (stack == 0xFFFFFFFFFFFFFFFFull)
            break;
        nodeHandle = n->parent;
    }
    *node_io = nodeHandle;
    *stack_io = stack;
    return 0;
}

int orc_scene_intersect(const orc_scene* s, int tri_mode, int svdag_lanes, orc_ray* r, orc_hit* h, orc_counters* c)
{
    uint32_t node = s->root, leaf = 0;
    uint64_t stack = 0xFFFFFFFFFFFFFFFFull;
    int hit = 0;
    if (h) {
        h->obj = ORC_MISS;
        h->inst = ORC_MISS;
        h->leaf = 0;
        h->u = h->v = 0.0f;
    }
    uint32_t parks = 0;
    while (orc_bvh4_step_ex(s, 0, svdag_lanes, r, &node, &stack, &leaf, c, parks < s->n_leaves)) {
        parks++;
        if (c)
            c->leaf_visits++;
        /* leaf task kernel, offline_batching...h:183-185: bottom level updates ray.tfar / si */
        if (orc_bvh8_intersect(&s->leaves[leaf].bvh, tri_mode, r, h, c)) {
            hit = 1;
            if (h)
                h->leaf = leaf;
        }
    }
    if (h)
        h->t = r->tfar;
    return hit;
}

int orc_scene_intersect_any(const orc_scene* s, int tri_mode, int svdag_lanes, orc_ray* r, orc_counters* c)
{
    uint32_t node = s->root, leaf = 0;
    uint64_t stack = 0xFFFFFFFFFFFFFFFFull;
    uint32_t parks = 0;
    while (orc_bvh4_step_ex(s, 1, svdag_lanes, r, &node, &stack, &leaf, c, parks < s->n_leaves)) {
        parks++;
        if (c)
            c->leaf_visits++;
        if (orc_bvh8_intersect_any(&s->leaves[leaf].bvh, tri_mode, r, c))
            return 1; /* :246-247 */
    }
    return 0;
}

/* ---------------------------------------------------------------- batch drivers */

typedef struct {
    int kind; /* 0 bvh8, 1 scene, 2 brute */
    const orc_bvh8* b;
    const orc_scene* s;
    const orc_mesh* m;
    size_t n_tris;
    int any_hit, tri_mode, lanes;
    const float* rays;
    size_t n, lo, hi;
    float *t, *u, *v;
    uint32_t *obj, *leaf, *anyhit, *visits, *inst;
    orc_counters c;
} job_t;

static void load_ray(const float* rays, size_t n, size_t i, orc_ray* r)
{
    r->o[0] = rays[0 * n + i];
    r->o[1] = rays[1 * n + i];
    r->o[2] = rays[2 * n + i];
    r->d[0] = rays[3 * n + i];
    r->d[1] = rays[4 * n + i];
    r->d[2] = rays[5 * n + i];
    r->tnear = rays[6 * n + i];
    r->tfar = rays[7 * n + i];
}

static void* job_run(void* p)
{
    job_t* j = (job_t*)p;
    memset(&j->c, 0, sizeof(j->c));
    for (size_t i = j->lo; i < j->hi; i++) {
        orc_ray r;
        load_ray(j->rays, j->n, i, &r);
        if (j->kind == 2) {
            uint32_t best = ORC_MISS;
            for (size_t k = 0; k < j->n_tris; k++) {
                float u, v;
                if (orc_tri_intersect(j->tri_mode, r.o, r.d, r.tnear, &r.tfar,
                        tri_vertex(j->m, (uint32_t)k, 0), tri_vertex(j->m, (uint32_t)k, 1), tri_vertex(j->m, (uint32_t)k, 2), &u, &v))
                    best = (uint32_t)k;
            }
            j->t[i] = r.tfar;
            j->obj[i] = best;
            continue;
        }
        if (j->any_hit) {
            uint64_t before = j->c.leaf_visits;
            int hit = j->kind == 0 ? orc_bvh8_intersect_any(j->b, j->tri_mode, &r, &j->c)
                                   : orc_scene_intersect_any(j->s, j->tri_mode, j->lanes, &r, &j->c);
            j->anyhit[i] = (uint32_t)hit;
            if (j->visits)
                j->visits[i] = (uint32_t)(j->c.leaf_visits - before);
        } else {
            orc_hit h;
            h.obj = ORC_MISS;
            h.inst = ORC_MISS;
            h.leaf = 0;
            h.u = h.v = 0.0f;
            uint64_t before = j->c.leaf_visits;
            if (j->kind == 0)
                orc_bvh8_intersect(j->b, j->tri_mode, &r, &h, &j->c);
            else
                orc_scene_intersect(j->s, j->tri_mode, j->lanes, &r, &h, &j->c);
            j->t[i] = r.tfar;
            j->obj[i] = h.obj;
            if (j->leaf)
                j->leaf[i] = h.leaf;
            j->u[i] = h.u;
            j->v[i] = h.v;
            if (j->inst)
                j->inst[i] = h.obj == ORC_MISS ? ORC_MISS : h.inst;
            if (j->visits)
                j->visits[i] = (uint32_t)(j->c.leaf_visits - before);
        }
    }
    return NULL;
}

static void run_jobs(job_t* proto, size_t n, int n_threads, orc_counters* c)
{
    if (n_threads < 1)
        n_threads = 1;
    if ((size_t)n_threads > n && n > 0)
        n_threads = (int)n;
    job_t* jobs = (job_t*)malloc(sizeof(job_t) * (size_t)n_threads);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int k = 0; k < n_threads; k++) {
        jobs[k] = *proto;
        jobs[k].lo = n * (size_t)k / (size_t)n_threads;
        jobs[k].hi = n * (size_t)(k + 1) / (size_t)n_threads;
        if (n_threads == 1)
            job_run(&jobs[k]);
        else
            pthread_create(&th[k], NULL, job_run, &jobs[k]);
    }
    if (c)
        memset(c, 0, sizeof(*c));
    for (int k = 0; k < n_threads; k++) {
        if (n_threads > 1)
            pthread_join(th[k], NULL);
        if (c) {
            c->svdag_tests += jobs[k].c.svdag_tests;
            c->svdag_culled += jobs[k].c.svdag_culled;
            c->leaf_visits += jobs[k].c.leaf_visits;
            c->inner_visits += jobs[k].c.inner_visits;
            c->leaf_pops += jobs[k].c.leaf_pops;
            c->tri_tests += jobs[k].c.tri_tests;
        }
    }
    free(jobs);
    free(th);
}

void orc_bvh8_trace(const orc_bvh8* b, int any_hit, int tri_mode, const float* rays, size_t n,
    float* t, uint32_t* obj, float* u, float* v, uint32_t* anyhit, orc_counters* c, int n_threads)
{
    orc_bvh8_trace_inst(b, any_hit, tri_mode, rays, n, t, obj, u, v, anyhit, NULL, c, n_threads);
}

void orc_bvh8_trace_inst(const orc_bvh8* b, int any_hit, int tri_mode, const float* rays, size_t n,
    float* t, uint32_t* obj, float* u, float* v, uint32_t* anyhit, uint32_t* inst, orc_counters* c, int n_threads)
{
    job_t j;
    memset(&j, 0, sizeof(j));
    j.kind = 0;
    j.b = b;
    j.any_hit = any_hit;
    j.tri_mode = tri_mode;
    j.rays = rays;
    j.n = n;
    j.t = t;
    j.obj = obj;
    j.u = u;
    j.v = v;
    j.anyhit = anyhit;
    j.inst = inst;
    run_jobs(&j, n, n_threads, c);
}

void orc_scene_trace(const orc_scene* s, int any_hit, int tri_mode, int svdag_lanes, const float* rays, size_t n,
    float* t, uint32_t* obj, uint32_t* leaf, float* u, float* v, uint32_t* anyhit, uint32_t* visits,
    orc_counters* c, int n_threads)
{
    orc_scene_trace_inst(s, any_hit, tri_mode, svdag_lanes, rays, n, t, obj, leaf, u, v, anyhit, visits, NULL, c, n_threads);
}

void orc_scene_trace_inst(const orc_scene* s, int any_hit, int tri_mode, int svdag_lanes, const float* rays, size_t n,
    float* t, uint32_t* obj, uint32_t* leaf, float* u, float* v, uint32_t* anyhit, uint32_t* visits, uint32_t* inst,
    orc_counters* c, int n_threads)
{
    job_t j;
    memset(&j, 0, sizeof(j));
    j.kind = 1;
    j.s = s;
    j.any_hit = any_hit;
    j.tri_mode = tri_mode;
    j.lanes = svdag_lanes;
    j.rays = rays;
    j.n = n;
    j.t = t;
    j.obj = obj;
    j.leaf = leaf;
    j.u = u;
    j.v = v;
    j.anyhit = anyhit;
    j.visits = visits;
    j.inst = inst;
    run_jobs(&j, n, n_threads, c);
}

void orc_brute_trace(const orc_mesh* m, size_t n_tris, int tri_mode, const float* rays, size_t n,
    float* t, uint32_t* prim, int n_threads)
{
    job_t j;
    memset(&j, 0, sizeof(j));
    j.kind = 2;
    j.m = m;
    j.n_tris = n_tris;
    j.tri_mode = tri_mode;
    j.rays = rays;
    j.n = n;
    j.t = t;
    j.obj = prim;
    run_jobs(&j, n, n_threads, NULL);
}

/* ---------------------------------------------------------------- film (sensor.cpp:36-49, :86-99) */
static uint64_t film_fixed(float c) { return (uint64_t)(int64_t)(c * 16777216.0f); } /* Pixel(color): scale by 2^24, truncate */

void orc_film_add(uint64_t* film, size_t n_pixels, const uint32_t* pixel, const float* r, const float* g, const float* b, size_t n)
{
    const float lim = 549755813888.0f; /* 2^39: beyond it the 40.24 format cannot hold the value (undefined in the reference) */
    for (size_t i = 0; i < n; ++i) {
        const float c[3] = { r[i], g[i], b[i] };
        /* :41-42: NaN / inf in any channel drops the contribution (every comparison with NaN is false) */
        if (!(pixel[i] < n_pixels && fabsf(c[0]) < lim && fabsf(c[1]) < lim && fabsf(c[2]) < lim))
            continue;
        for (int k = 0; k < 3; ++k)
            film[(size_t)pixel[i] * 3 + (size_t)k] += film_fixed(c[k]); /* :44-48: the CAS loop is an atomic add */
    }
}

void orc_film_resolve(const uint64_t* film, size_t n_pixels, float* rgb_out)
{
    for (size_t i = 0; i < n_pixels * 3; ++i)
        rgb_out[i] = (float)film[i] * 5.9604644775390625e-08f;
}

void orc_svdag_trace(const orc_svdag* d, int lanes, const float* rays, size_t n, unsigned char* out)
{
    for (size_t i = 0; i < n; ++i) {
        orc_ray r;
        r.o[0] = rays[0 * n + i];
        r.o[1] = rays[1 * n + i];
        r.o[2] = rays[2 * n + i];
        r.d[0] = rays[3 * n + i];
        r.d[1] = rays[4 * n + i];
        r.d[2] = rays[5 * n + i];
        r.tnear = rays[6 * n + i];
        r.tfar = rays[7 * n + i];
        out[i] = (unsigned char)orc_svdag_intersect(d, &r, lanes);
    }
}

/* ---------------------------------------------------------------- benchmark integrator pieces (N4) */

/* samplers/rng/pcg.cpp:32-39 */
static uint32_t pcg_u32(uint64_t* state, uint64_t inc)
{
    uint64_t old = *state;
    *state = old * 0x5851F42D4C957F2DULL + inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
/* :23-30 */
static void pcg_seed(uint64_t initseq, uint64_t* state, uint64_t* inc)
{
    *state = 0;
    *inc = (initseq << 1u) | 1u;
    pcg_u32(state, *inc);
    *state += 0x853C49E6748FEA9BULL;
    pcg_u32(state, *inc);
}
/* :58-61 */
static float pcg_float(uint64_t* state, uint64_t inc)
{
    float f = (float)pcg_u32(state, inc) * 0x1p-32f;
    float lim = 1.0f - FLT_EPSILON;
    return lim < f ? lim : f; /* std::min(lim, f) */
}

static void normalize3(float v[3])
{
    /* glm::normalize = v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x) */
    float inv = 1.0f / sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    v[0] *= inv;
    v[1] *= inv;
    v[2] *= inv;
}

void orc_camera_rays(const orc_camera* cam, uint32_t width, uint32_t height, uint32_t spp, uint64_t seed, uint64_t first, size_t n,
    float* rays, const orc_paths* paths)
{
    const float resx = (float)width, resy = (float)height;
    for (size_t k = 0; k < n; k++) {
        const uint64_t i = first + k; /* sampler_integrator.cpp:145-160 */
        const uint64_t pixelIndex = i / spp;
        const uint32_t x = (uint32_t)(pixelIndex % width), y = (uint32_t)(pixelIndex / width);
        uint64_t st, inc;
        pcg_seed(seed + i, &st, &inc);
        const float r0 = pcg_float(&st, inc), r1 = pcg_float(&st, inc);
        const float sx = ((float)x + r0) / resx, sy = ((float)y + r1) / resy;
        const float cx = sx - 0.5f, cy = sy - 0.5f; /* perspective_camera.cpp:39 */
        float o[3], d[3];
        for (int a = 0; a < 3; a++) {
            o[a] = (cam->lens_center[a] + cx * cam->lens_du[a]) + cy * cam->lens_dv[a];
            const float sc = (cam->screen_center[a] + cx * cam->screen_du[a]) + cy * cam->screen_dv[a];
            d[a] = sc - o[a];
        }
        normalize3(d);
        for (int a = 0; a < 3; a++) {
            rays[(size_t)a * n + k] = o[a];
            rays[(size_t)(3 + a) * n + k] = d[a];
        }
        rays[6 * n + k] = 0.0f;
        rays[7 * n + k] = FLT_MAX;
        if (paths) {
            paths->rng_state[k] = st;
            paths->rng_inc[k] = inc;
            paths->pixel[k] = y * width + x;
            paths->wr[k] = paths->wg[k] = paths->wb[k] = 1.0f;
        }
        (void)height;
    }
}

/* a path that ended (miss: path_integrator.cpp:113-127, failed bounce: :109-112): unhittable ray, weight zero -- it contributes nothing more */
static void dead_ray(float* rays, const orc_paths* paths, size_t n, size_t k)
{
    paths->wr[k] = paths->wg[k] = paths->wb[k] = 0.0f;
    for (int a = 0; a < 6; a++)
        rays[(size_t)a * n + k] = a == 3 ? 1.0f : 0.0f;
    rays[6 * n + k] = 1.0f;
    rays[7 * n + k] = 0.0f;
}

void orc_lambert_bounce(const orc_bvh8* b1, const orc_scene* s, const float* rays_in, const float* t, const uint32_t* obj, const uint32_t* leaf,
    const float* u, const float* v, size_t n, const float albedo[3], float* rays, const orc_paths* paths)
{
    (void)t;
    for (size_t k = 0; k < n; k++) {
        if (obj[k] == ORC_MISS) {
            dead_ray(rays, paths, n, k);
            continue;
        }
        const orc_bvh8* b = b1 ? b1 : &s->leaves[leaf[k]].bvh;
        const uint32_t prim = b->obj_prim ? b->obj_prim[obj[k]] : obj[k];
        const float *p0 = tri_vertex(&b->mesh, prim, 0), *p1 = tri_vertex(&b->mesh, prim, 1), *p2 = tri_vertex(&b->mesh, prim, 2);
        const float dir[3] = { rays_in[3 * n + k], rays_in[4 * n + k], rays_in[5 * n + k] };
        /* triangle_intersect.cpp:458-460 */
        float e1[3], e2[3], ng[3];
        for (int a = 0; a < 3; a++) {
            e1[a] = p1[a] - p0[a];
            e2[a] = p2[a] - p0[a];
        }
        cross3(e1, e2, ng);
        normalize3(ng);
        if (((ng[0] * -dir[0] + ng[1] * -dir[1]) + ng[2] * -dir[2]) < 0.0f)
            for (int a = 0; a < 3; a++)
                ng[a] = -ng[a];
        /* triangle_shading.cpp:14-21 */
        const float b0 = 1.0f - u[k] - v[k], bb1 = u[k], bb2 = v[k];
        float pos[3];
        for (int a = 0; a < 3; a++)
            pos[a] = (b0 * p0[a] + bb1 * p1[a]) + bb2 * p2[a];
        /* material.cpp:5-20 */
        const float helper[3] = { ng[0] > 0.5f ? 0.0f : 1.0f, ng[0] > 0.5f ? 1.0f : 0.0f, 0.0f };
        float ss[3], ts[3];
        cross3(ng, helper, ss);
        normalize3(ss);
        cross3(ng, ss, ts);
        normalize3(ts);
        /* path_integrator.cpp:142 -> material.cpp:85-100 -> LambertianReflection::sampleF -> sampling.h:8-49 */
        uint64_t st = paths->rng_state[k];
        const uint64_t inc = paths->rng_inc[k];
        const float r0 = pcg_float(&st, inc), r1 = pcg_float(&st, inc);
        const float ox = 2.0f * r0 - 1.0f, oy = 2.0f * r1 - 1.0f;
        float dx = 0.0f, dy = 0.0f;
        if (!(ox == 0.0f && oy == 0.0f)) {
            const float pi = 3.14159265358979323846264338327950288f;
            float theta, r;
            if (fabsf(ox) > fabsf(oy)) {
                r = ox;
                theta = pi / 4.0f * (oy / ox);
            } else {
                r = oy;
                theta = pi / 2.0f - pi / 4.0f * (ox / oy);
            }
            dx = r * cosf(theta);
            dy = r * sinf(theta);
        }
        const float zz = 1.0f - dx * dx - dy * dy;
        float wl[3] = { dx, dy, sqrtf(0.0f > zz ? 0.0f : zz) }; /* std::max(0.0f, zz) */
        /* wo = -dir faces the flipped normal, so wo.z = dot(wo, ns) >= 0: LambertianReflection keeps wi.z unless wo.z < 0 */
        const float woz = (-dir[0] * ng[0] + -dir[1] * ng[1]) + -dir[2] * ng[2];
        if (woz < 0.0f)
            wl[2] = -wl[2];
        float wi[3];
        for (int a = 0; a < 3; a++)
            wi[a] = (ss[a] * wl[0] + ts[a] * wl[1]) + ng[a] * wl[2]; /* material.cpp:41-47 */
        /* pdf = |wi.z| / pi when wo, wi lie in one hemisphere, else 0 -> no bounce (material.cpp:93-94, path_integrator.cpp:142) */
        const float inv_pi = 0.318309886183790671537767526745028724f;
        const float pdf = (woz * wl[2] > 0.0f) ? fabsf(wl[2]) * inv_pi : 0.0f;
        /* rayState.rng = PcgRng(rng.uniformU64()) (:151) -- drawn whether or not the bounce survives keeps the stream position simple */
        const uint64_t left = pcg_u32(&st, inc), right = pcg_u32(&st, inc);
        uint64_t nst, ninc;
        pcg_seed((left << 32) | right, &nst, &ninc);
        paths->rng_state[k] = nst;
        paths->rng_inc[k] = ninc;
        if (!(pdf > 0.0f) || (albedo[0] == 0.0f && albedo[1] == 0.0f && albedo[2] == 0.0f)) {
            dead_ray(rays, paths, n, k);
            continue;
        }
        /* weight = prev * f * absDot(wi, ns) / pdf with f = albedo / pi (:152) */
        const float adot = fabsf((wi[0] * ng[0] + wi[1] * ng[1]) + wi[2] * ng[2]);
        paths->wr[k] = paths->wr[k] * (albedo[0] * inv_pi) * adot / pdf;
        paths->wg[k] = paths->wg[k] * (albedo[1] * inv_pi) * adot / pdf;
        paths->wb[k] = paths->wb[k] * (albedo[2] * inv_pi) * adot / pdf;
        /* interaction.cpp:39-55: origin offset along the normal on the side the ray leaves from */
        const float side = ((ng[0] * wi[0] + ng[1] * wi[1]) + ng[2] * wi[2]) > 0.0f ? 1.0f : -1.0f;
        for (int a = 0; a < 3; a++) {
            rays[(size_t)a * n + k] = side > 0.0f ? pos[a] + ng[a] * 0.000005f : pos[a] - ng[a] * 0.000005f;
            rays[(size_t)(3 + a) * n + k] = wi[a];
        }
        rays[6 * n + k] = 0.0f;
        rays[7 * n + k] = FLT_MAX;
    }
}

void orc_splat_misses(uint64_t* film, size_t n_pixels, const uint32_t* obj, const orc_paths* paths, size_t n, const float sky[3])
{
    for (size_t k = 0; k < n; k++) {
        if (obj[k] != ORC_MISS)
            continue;
        /* path_integrator.cpp:113-125: weight * Le; sensor.cpp:36-49 */
        const float r = paths->wr[k] * sky[0], g = paths->wg[k] * sky[1], b = paths->wb[k] * sky[2];
        orc_film_add(film, n_pixels, &paths->pixel[k], &r, &g, &b, 1);
    }
}
