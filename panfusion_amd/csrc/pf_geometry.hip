// Spherical geometry kernels for gfx950: e2p / p2e sampling grids, grid_sample-compatible remap,
// EPA correspondence-bias tables and the spherical positional encoding.
//
// Arithmetic follows the reference literally (float64 ray math on float32-rounded rotation
// matrices, float32 grid_sample position round trip); this file is compiled with
// -ffp-contract=off so that no multiply-add is fused where numpy / torch round twice.
//
// Reference: external/Perspective_and_Equirectangular/e2p.py:9-76, p2e.py:9-71,
// models/pano/utils.py:10-106, models/modules/transformer.py:165-201.
#include "pf_common.h"
#include <math.h>
#include <string.h>
#include <vector>

namespace pf {

static constexpr double kPi = 3.14159265358979323846;
static constexpr double kDeg2Rad = kPi / 180.0;   // numpy: x * (NPY_PI / 180.0)

struct CamParams {
    double R1[9], R2[9];      // float32-rounded values held in double (np.dot promotes them)
    double R1i[9], R2i[9];    // float32-rounded inverses (np.linalg.inv on float32 input)
    double w_len, h_len;
};

// ---- host: per-camera constants ------------------------------------------------------------
static void rodrigues_f32(const float rvec[3], double R[9]) {
    // OpenCV: compute in double from the float32 vector, convert the matrix back to float32.
    double r0 = rvec[0], r1 = rvec[1], r2 = rvec[2];
    double theta = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
    double M[9];
    if (theta < 2.220446049250313e-16) {
        for (int i = 0; i < 9; ++i) M[i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else {
        double c = cos(theta), s = sin(theta), c1 = 1.0 - c, it = 1.0 / theta;
        double x = r0 * it, y = r1 * it, z = r2 * it;
        double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
        double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
        for (int i = 0; i < 9; ++i) M[i] = (c * ((i % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[i]) + s * rx[i];
    }
    for (int i = 0; i < 9; ++i) R[i] = static_cast<double>(static_cast<float>(M[i]));
}

static void inverse_f32(const double A[9], double Ai[9]) {
    // np.linalg.inv on a float32 matrix (p2e.py:28-29): numpy's `inv` always computes in DOUBLE (gufunc
    // signature 'd->d': LAPACK dgesv, LU with partial pivoting on [A | I]) and casts the result to the input
    // type.  Same here: LU with partial pivoting in double, then one rounding to float32 -- the double result
    // is ~1e-16 from the exact inverse, so the float32 value is the correctly rounded inverse in all but
    // measure-zero cases, whatever the operation order inside LAPACK.
    double M[3][6];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { M[r][c] = A[3 * r + c]; M[r][3 + c] = (r == c) ? 1.0 : 0.0; }
    for (int j = 0; j < 3; ++j) {
        int p = j;
        for (int r = j + 1; r < 3; ++r) if (fabs(M[r][j]) > fabs(M[p][j])) p = r;
        if (p != j) for (int c = 0; c < 6; ++c) { double t = M[j][c]; M[j][c] = M[p][c]; M[p][c] = t; }
        for (int r = j + 1; r < 3; ++r) {
            const double l = M[r][j] / M[j][j];
            for (int c = j; c < 6; ++c) M[r][c] -= l * M[j][c];
        }
    }
    for (int c = 3; c < 6; ++c)
        for (int r = 2; r >= 0; --r) {
            double v = M[r][c];
            for (int k = r + 1; k < 3; ++k) v -= M[r][k] * M[k][c];
            M[r][c] = v / M[r][r];
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Ai[3 * r + c] = static_cast<double>(static_cast<float>(M[r][3 + c]));
}

static void make_cam(double fov, double theta, double phi, int vh, int vw, CamParams* cp) {
    // e2p.py:10-13 / p2e.py:10-13
    double hfov = static_cast<double>(vh) / vw * fov;
    cp->w_len = tan((fov / 2.0) * kDeg2Rad);
    cp->h_len = tan((hfov / 2.0) * kDeg2Rad);
    // e2p.py:23-26 under numpy 1.26 promotion: float32 axis * float32(scalar)
    float yaw = static_cast<float>(theta * kDeg2Rad);
    float pitch = static_cast<float>((-phi) * kDeg2Rad);
    float rv1[3] = {0.0f * yaw, 0.0f * yaw, 1.0f * yaw};
    rodrigues_f32(rv1, cp->R1);
    // np.dot(R1, y_axis) in float32 = second column of R1
    float ax[3] = {static_cast<float>(cp->R1[1]), static_cast<float>(cp->R1[4]), static_cast<float>(cp->R1[7])};
    float rv2[3] = {ax[0] * pitch, ax[1] * pitch, ax[2] * pitch};
    rodrigues_f32(rv2, cp->R2);
    inverse_f32(cp->R1, cp->R1i);
    inverse_f32(cp->R2, cp->R2i);
}

// ---- device helpers ------------------------------------------------------------------------
__device__ __forceinline__ double linspace_at(double start, double stop, int num, int i) {
    // numpy.linspace(endpoint=True): arange*step + start, last element forced to stop.
    if (i == num - 1 && num > 1) return stop;
    double step = (stop - start) / static_cast<double>(num - 1);
    return static_cast<double>(i) * step + start;
}

__device__ __forceinline__ void matvec(const double* R, double x, double y, double z,
                                       double& ox, double& oy, double& oz) {
    ox = (R[0] * x + R[1] * y) + R[2] * z;
    oy = (R[3] * x + R[4] * y) + R[5] * z;
    oz = (R[6] * x + R[7] * y) + R[8] * z;
}

// e2p.py:9-36: lon/lat (radians) of view pixel (y, x); lat negative towards the top row.
__device__ __forceinline__ void pers_pixel_lonlat(const CamParams& cp, int h, int w, int y, int x,
                                                  double& lon, double& lat) {
    double ry = linspace_at(-cp.w_len, cp.w_len, w, x);
    double rz = -linspace_at(-cp.h_len, cp.h_len, h, y);
    double d = sqrt((1.0 + ry * ry) + rz * rz);
    double vx = 1.0 / d, vy = ry / d, vz = rz / d;
    double ax, ay, az, bx, by, bz;
    matvec(cp.R1, vx, vy, vz, ax, ay, az);
    matvec(cp.R2, ax, ay, az, bx, by, bz);
    lat = -asin(bz);
    lon = atan2(by, bx);
}

// e2p.py:39-51
__device__ __forceinline__ void e2p_position(const CamParams& cp, int eh, int ew, int h, int w,
                                             int y, int x, double& px, double& py,
                                             double& lon, double& lat) {
    pers_pixel_lonlat(cp, h, w, y, x, lon, lat);
    double cx = (ew - 1) / 2.0, cy = (eh - 1) / 2.0;
    double lo = lon / kPi * 180.0, la = lat / kPi * 180.0;
    px = lo / 180.0 * cx + cx;
    py = la / 90.0 * cy + cy;
}

// p2e.py:9-49
__device__ __forceinline__ void p2e_position(const CamParams& cp, int ph, int pw, int h, int w,
                                             int y, int x, double& u, double& v, bool& visible) {
    double lon = linspace_at(-180.0, 180.0, w, x) * kDeg2Rad;
    double lat = linspace_at(90.0, -90.0, h, y) * kDeg2Rad;
    double cl = cos(lat);
    double dx = cos(lon) * cl, dy = sin(lon) * cl, dz = sin(lat);
    double ax, ay, az, bx, by, bz;
    matvec(cp.R2i, dx, dy, dz, ax, ay, az);
    matvec(cp.R1i, ax, ay, az, bx, by, bz);
    bool front = bx > 0.0;
    double yy = by / bx, zz = bz / bx;
    bool inside = (-cp.w_len < yy) && (yy < cp.w_len) && (-cp.h_len < zz) && (zz < cp.h_len);
    u = inside ? (yy + cp.w_len) / 2.0 / cp.w_len * pw : 0.0;
    v = inside ? (-zz + cp.h_len) / 2.0 / cp.h_len * ph : 0.0;
    visible = inside && front;
}

// kornia.remap normalisation (factor first) + torch grid_sample un-normalisation, fp32.
__device__ __forceinline__ float sample_position(float coord, int size) {
    float factor = 2.0f / static_cast<float>(size - 1);
    float xn = factor * coord - 1.0f;
    return ((xn + 1.0f) / 2.0f) * static_cast<float>(size - 1);
}

// ---- grids -----------------------------------------------------------------------------------
// Camera constants travel BY VALUE in the kernel arguments (<= CAM_BATCH cameras per launch, 3.6 KB of the
// 4 KB argument block): no host-to-device copy of a host temporary, hence no stream synchronisation and no
// library-owned device scratch (include/panfusion_hip.h: "no implicit device sync").
constexpr int CAM_BATCH = 12;
struct CamBatch { CamParams c[CAM_BATCH]; };

__global__ void k_store_cams(const CamBatch batch, int n, CamParams* dst) {      // workspace copy for the table builder
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int WORDS = sizeof(CamParams) / sizeof(double);
    if (i < n * WORDS) reinterpret_cast<double*>(dst)[i] = reinterpret_cast<const double*>(batch.c)[i];
}

__global__ void k_e2p_grid(const CamBatch cams, int ncam, int eh, int ew, int h, int w,
                           float* map_x, float* map_y, float* lonlat) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    long total = static_cast<long>(ncam) * h * w;
    if (i >= total) return;
    int x = i % w, y = (i / w) % h, c = i / (static_cast<long>(w) * h);
    double px, py, lon, lat;
    e2p_position(cams.c[c], eh, ew, h, w, y, x, px, py, lon, lat);
    if (map_x) map_x[i] = static_cast<float>(px);
    if (map_y) map_y[i] = static_cast<float>(py);
    if (lonlat) {
        lonlat[2 * i] = static_cast<float>(lon);
        lonlat[2 * i + 1] = static_cast<float>(lat);
    }
}

__global__ void k_p2e_grid(const CamBatch cams, int ncam, int ph, int pw, int h, int w,
                           float* map_u, float* map_v, uint8_t* mask) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    long total = static_cast<long>(ncam) * h * w;
    if (i >= total) return;
    int x = i % w, y = (i / w) % h, c = i / (static_cast<long>(w) * h);
    double u, v;
    bool vis;
    p2e_position(cams.c[c], ph, pw, h, w, y, x, u, v, vis);
    map_u[i] = static_cast<float>(u);
    map_v[i] = static_cast<float>(v);
    mask[i] = vis ? 1 : 0;
}

__global__ void k_nearest_indices(const float* map_x, const float* map_y, long n, int sh, int sw,
                                  int32_t* idx) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    float fx = nearbyintf(sample_position(map_x[i], sw));
    float fy = nearbyintf(sample_position(map_y[i], sh));
    bool ok = fx >= 0.0f && fx < static_cast<float>(sw) && fy >= 0.0f && fy < static_cast<float>(sh);
    idx[i] = ok ? static_cast<int>(fy) * sw + static_cast<int>(fx) : -1;
}

__global__ void k_equi_coords(int H, int W, float* lonlat) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    int x = i % W, y = i / W;
    lonlat[2 * i] = static_cast<float>(linspace_at(-kPi, kPi, W, x));
    lonlat[2 * i + 1] = static_cast<float>(linspace_at(kPi / 2, -kPi / 2, H, y));
}

__global__ void k_spherical_pe(const float* coords, long n, const float* freq, int nf, float* out) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= n * nf) return;
    long row = i / nf;
    int f = i % nf;
    float a = coords[2 * row] * freq[f];
    float b = coords[2 * row + 1] * freq[f];
    float* o = out + row * 4 * nf;
    o[f] = sinf(a);
    o[nf + f] = sinf(b);
    o[2 * nf + f] = cosf(a);
    o[3 * nf + f] = cosf(b);
}

// ---- remap -------------------------------------------------------------------------------------
template <typename S> __device__ __forceinline__ float load_as_f32(const S* p);
template <> __device__ __forceinline__ float load_as_f32<float>(const float* p) { return *p; }
struct RawBf16 { unsigned short v; };
struct RawF16 { unsigned short v; };
template <> __device__ __forceinline__ float load_as_f32<RawBf16>(const RawBf16* p) { return to_f32<Bf16>(p->v); }
template <> __device__ __forceinline__ float load_as_f32<RawF16>(const RawF16* p) { return to_f32<F16>(p->v); }
__device__ __forceinline__ void store_from_f32(float* p, float v) { *p = v; }
__device__ __forceinline__ void store_from_f32(RawBf16* p, float v) { p->v = from_f32<Bf16>(v); }
__device__ __forceinline__ void store_from_f32(RawF16* p, float v) { p->v = from_f32<F16>(v); }

template <typename S>
__global__ void k_remap(const S* src, int B, int C, int hs, int ws, const float* map_x,
                        const float* map_y, const uint8_t* mask, int map_batch, int ho, int wo,
                        int mode, S* dst) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    long per = static_cast<long>(ho) * wo;
    if (i >= B * per) return;
    int b = i / per;
    long pix = i % per;
    long mi = (map_batch == 1 ? 0 : b) * per + pix;
    float px = sample_position(map_x[mi], ws);
    float py = sample_position(map_y[mi], hs);
    float mk = mask ? static_cast<float>(mask[mi]) : 1.0f;
    const S* sb = src + static_cast<long>(b) * C * hs * ws;
    S* db = dst + static_cast<long>(b) * C * per + pix;
    long cs = static_cast<long>(hs) * ws;
    if (mode == 0) {
        float fx = nearbyintf(px), fy = nearbyintf(py);
        bool ok = fx >= 0.0f && fx < static_cast<float>(ws) && fy >= 0.0f && fy < static_cast<float>(hs);
        long off = ok ? static_cast<long>(fy) * ws + static_cast<long>(fx) : 0;
        for (int c = 0; c < C; ++c) {
            float v = ok ? load_as_f32(sb + c * cs + off) : 0.0f;
            store_from_f32(db + c * per, mask ? v * mk : v);
        }
    } else {
        float x0f = floorf(px), y0f = floorf(py);
        float wx = px - x0f, ex = 1.0f - wx, wy = py - y0f, sy = 1.0f - wy;
        float w_nw = sy * ex, w_ne = sy * wx, w_sw = wy * ex, w_se = wy * wx;
        int x0 = static_cast<int>(x0f), y0 = static_cast<int>(y0f);
        bool xin0 = x0 >= 0 && x0 < ws, xin1 = x0 + 1 >= 0 && x0 + 1 < ws;
        bool yin0 = y0 >= 0 && y0 < hs, yin1 = y0 + 1 >= 0 && y0 + 1 < hs;
        for (int c = 0; c < C; ++c) {
            const S* sc = sb + c * cs;
            float acc = 0.0f;
            if (yin0 && xin0) acc = load_as_f32(sc + static_cast<long>(y0) * ws + x0) * w_nw;
            if (yin0 && xin1) acc += load_as_f32(sc + static_cast<long>(y0) * ws + x0 + 1) * w_ne;
            if (yin1 && xin0) acc += load_as_f32(sc + static_cast<long>(y0 + 1) * ws + x0) * w_sw;
            if (yin1 && xin1) acc += load_as_f32(sc + static_cast<long>(y0 + 1) * ws + x0 + 1) * w_se;
            store_from_f32(db + c * per, mask ? acc * mk : acc);
        }
    }
}

// ---- EPA tables ------------------------------------------------------------------------------
// Dense scratch W1[v][e][p] = bilinear weight of view pixel p in the p2e sample of pano pixel e
// (models/pano/utils.py:31-34), W2[v][e][p] = weight of pano pixel e in the e2p sample of view
// pixel p (:35-38).  Each (e,p) receives at most one contribution from each map.
__global__ void k_scatter_p2e(const CamParams* cams, int m, int ph, int pw, int eh, int ew, float* W1) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    long E = static_cast<long>(eh) * ew, P = static_cast<long>(ph) * pw;
    if (i >= m * E) return;
    int v = i / E;
    long e = i % E;
    int ex_ = e % ew, ey_ = e / ew;
    double u, vv;
    bool vis;
    p2e_position(cams[v], ph, pw, eh, ew, ey_, ex_, u, vv, vis);
    if (!vis) return;
    float px = sample_position(static_cast<float>(u), pw);
    float py = sample_position(static_cast<float>(vv), ph);
    float x0f = floorf(px), y0f = floorf(py);
    float wx = px - x0f, ex = 1.0f - wx, wy = py - y0f, sy = 1.0f - wy;
    int x0 = static_cast<int>(x0f), y0 = static_cast<int>(y0f);
    float* row = W1 + (static_cast<long>(v) * E + e) * P;
    const float wgt[4] = {sy * ex, sy * wx, wy * ex, wy * wx};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int xc = x0 + (k & 1), yc = y0 + (k >> 1);
        if (xc >= 0 && xc < pw && yc >= 0 && yc < ph) row[static_cast<long>(yc) * pw + xc] = wgt[k];
    }
}

__global__ void k_scatter_e2p(const CamParams* cams, int m, int ph, int pw, int eh, int ew, float* W2) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    long E = static_cast<long>(eh) * ew, P = static_cast<long>(ph) * pw;
    if (i >= m * P) return;
    int v = i / P;
    long p = i % P;
    int x = p % pw, y = p / pw;
    double dpx, dpy, lon, lat;
    e2p_position(cams[v], eh, ew, ph, pw, y, x, dpx, dpy, lon, lat);
    float px = sample_position(static_cast<float>(dpx), ew);
    float py = sample_position(static_cast<float>(dpy), eh);
    float x0f = floorf(px), y0f = floorf(py);
    float wx = px - x0f, ex = 1.0f - wx, wy = py - y0f, sy = 1.0f - wy;
    int x0 = static_cast<int>(x0f), y0 = static_cast<int>(y0f);
    const float wgt[4] = {sy * ex, sy * wx, wy * ex, wy * wx};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int xc = x0 + (k & 1), yc = y0 + (k >> 1);
        if (xc >= 0 && xc < ew && yc >= 0 && yc < eh)
            W2[(static_cast<long>(v) * E + static_cast<long>(yc) * ew + xc) * P + p] = wgt[k];
    }
}

// utils.py:49-56: A = clamp(W2 + W1) (in place in W2); B[v][p][e] = clamp(W1 + A).
__global__ void k_crossfill(const float* W1, float* W2, float* Bt, int m, long E, long P) {
    long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= m * E * P) return;
    float w1 = W1[i];
    float a = fminf(fmaxf(W2[i] + w1, 0.0f), 1.0f);
    W2[i] = a;
    float b = fminf(fmaxf(w1 + a, 0.0f), 1.0f);
    if (b != 0.0f) {
        long p = i % P, e = (i / P) % E, v = i / (P * E);
        Bt[(v * P + p) * E + e] = b;
    }
}

// utils.py:61-76 for one image per block: 5x5 separable Gaussian (horizontal then vertical,
// replicate border; circular in x for the panorama side, which the reference pads by 2 first),
// divide by the image maximum, times 2.  Output is bias+1; all-zero images are skipped (their
// rows of the pre-zeroed table stay 0).
template <bool PANO_ROWS>
__global__ void k_blur_normalise(const float* src, int rows_per_view, int h, int w, float g0,
                                 float g1, float g2, float* table, long q_stride_or_E, long mP,
                                 uint8_t* flags, int flags_ld) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* img = sm;
    float* tmp = sm + h * w;
    __shared__ float red[8];
    const int n = h * w;
    const long row = blockIdx.x;
    const float* s = src + row * n;
    float any = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float v = s[i];
        img[i] = v;
        any = fmaxf(any, v);
    }
    // block max (values are >= 0)
    for (int o = 32; o > 0; o >>= 1) any = fmaxf(any, __shfl_xor(any, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = any;
    __syncthreads();
    float tot = 0.0f;
    for (int i = 0; i < (blockDim.x >> 6); ++i) tot = fmaxf(tot, red[i]);
    if (tot == 0.0f) return;
    const float g[5] = {g0, g1, g2, g1, g0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int x = i % w, y = i / w;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int xx = x + k - 2;
            if (PANO_ROWS) xx = (xx + w) % w; else xx = min(max(xx, 0), w - 1);
            acc += g[k] * img[y * w + xx];
        }
        tmp[i] = acc;
    }
    __syncthreads();
    float peak = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int x = i % w, y = i / w;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int yy = min(max(y + k - 2, 0), h - 1);
            acc += g[k] * tmp[yy * w + x];
        }
        img[i] = acc;
        peak = fmaxf(peak, acc);
    }
    for (int o = 32; o > 0; o >>= 1) peak = fmaxf(peak, __shfl_xor(peak, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = peak;
    __syncthreads();
    peak = 0.0f;
    for (int i = 0; i < (blockDim.x >> 6); ++i) peak = fmaxf(peak, red[i]);
    if (peak == 0.0f) peak = 1.0f;
    long v = row / rows_per_view, r = row % rows_per_view;
    long q, kbase;
    float* out;
    if (PANO_ROWS) {            // src rows (v, p) over pano pixels: table bias_p[(v*P+p)][E]
        q = v * rows_per_view + r;
        kbase = 0;
        out = table + q * q_stride_or_E;
    } else {                    // src rows (v, e) over view pixels: table bias_e[e][v*P + :]
        q = r;
        kbase = v * n;
        out = table + q * mP + kbase;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float val = img[i] / peak * 2.0f;
        out[i] = val;
        if (val != 0.0f) flags[(q >> 5) * flags_ld + ((kbase + i) >> 5)] = 1;
    }
}

// Host side of the by-value camera batches: cameras [c0, c0 + n) of the call.
static void fill_batch(const double* fov, const double* theta, const double* phi, int c0, int n, int vh, int vw,
                       CamBatch* b) {
    for (int i = 0; i < n; ++i) make_cam(fov[c0 + i], theta[c0 + i], phi[c0 + i], vh, vw, &b->c[i]);
    for (int i = n; i < CAM_BATCH; ++i) b->c[i] = b->c[0];
}

}  // namespace pf

using namespace pf;

extern "C" pf_status pf_e2p_grid(const double* fov, const double* theta, const double* phi, int ncam,
                                 int eh, int ew, int h, int w, float* map_x, float* map_y,
                                 float* lonlat, void* stream) {
    PF_REQUIRE(fov && theta && phi && ncam > 0, "pf_e2p_grid: cameras missing");
    PF_REQUIRE(eh > 1 && ew > 1 && h > 1 && w > 1, "pf_e2p_grid: sizes must be > 1");
    hipStream_t st = as_stream(stream);
    const long per = static_cast<long>(h) * w;
    for (int c0 = 0; c0 < ncam; c0 += CAM_BATCH) {
        const int n = ncam - c0 < CAM_BATCH ? ncam - c0 : CAM_BATCH;
        CamBatch batch;
        fill_batch(fov, theta, phi, c0, n, h, w, &batch);
        hipLaunchKernelGGL(k_e2p_grid, dim3(cdiv(n * per, 256)), dim3(256), 0, st, batch, n, eh, ew, h, w,
                           map_x ? map_x + c0 * per : nullptr, map_y ? map_y + c0 * per : nullptr,
                           lonlat ? lonlat + 2 * c0 * per : nullptr);
    }
    PF_CHECK_LAUNCH("pf_e2p_grid");
    return PF_OK;
}

extern "C" pf_status pf_p2e_grid(const double* fov, const double* theta, const double* phi, int ncam,
                                 int ph, int pw, int h, int w, float* map_u, float* map_v,
                                 uint8_t* mask, void* stream) {
    PF_REQUIRE(fov && theta && phi && ncam > 0, "pf_p2e_grid: cameras missing");
    PF_REQUIRE(ph > 1 && pw > 1 && h > 1 && w > 1, "pf_p2e_grid: sizes must be > 1");
    PF_REQUIRE(map_u && map_v && mask, "pf_p2e_grid: outputs missing");
    hipStream_t st = as_stream(stream);
    const long per = static_cast<long>(h) * w;
    for (int c0 = 0; c0 < ncam; c0 += CAM_BATCH) {
        const int n = ncam - c0 < CAM_BATCH ? ncam - c0 : CAM_BATCH;
        CamBatch batch;
        fill_batch(fov, theta, phi, c0, n, ph, pw, &batch);
        hipLaunchKernelGGL(k_p2e_grid, dim3(cdiv(n * per, 256)), dim3(256), 0, st, batch, n, ph, pw, h, w,
                           map_u + c0 * per, map_v + c0 * per, mask + c0 * per);
    }
    PF_CHECK_LAUNCH("pf_p2e_grid");
    return PF_OK;
}

extern "C" pf_status pf_nearest_indices(const float* map_x, const float* map_y, long n, int sh, int sw,
                                        int32_t* idx, void* stream) {
    PF_REQUIRE(map_x && map_y && idx && n > 0 && sh > 1 && sw > 1, "pf_nearest_indices: bad arguments");
    hipLaunchKernelGGL(k_nearest_indices, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), map_x,
                       map_y, n, sh, sw, idx);
    PF_CHECK_LAUNCH("pf_nearest_indices");
    return PF_OK;
}

extern "C" pf_status pf_remap(const void* src, int dtype, int B, int C, int hs, int ws,
                              const float* map_x, const float* map_y, const uint8_t* mask,
                              int map_batch, int ho, int wo, int mode, void* dst, void* stream) {
    PF_REQUIRE(src && dst && map_x && map_y, "pf_remap: null pointer");
    PF_REQUIRE(B > 0 && C > 0 && hs > 1 && ws > 1 && ho > 0 && wo > 0, "pf_remap: bad sizes");
    PF_REQUIRE(map_batch == 1 || map_batch == B, "pf_remap: map_batch must be 1 or B");
    PF_REQUIRE(mode == 0 || mode == 1, "pf_remap: mode must be 0 (nearest) or 1 (bilinear)");
    long total = static_cast<long>(B) * ho * wo;
    dim3 grid(cdiv(total, 256)), block(256);
    hipStream_t st = as_stream(stream);
    if (dtype == PF_F32)
        hipLaunchKernelGGL(k_remap<float>, grid, block, 0, st, static_cast<const float*>(src), B, C, hs, ws,
                           map_x, map_y, mask, map_batch, ho, wo, mode, static_cast<float*>(dst));
    else if (dtype == PF_BF16)
        hipLaunchKernelGGL(k_remap<RawBf16>, grid, block, 0, st, static_cast<const RawBf16*>(src), B, C, hs,
                           ws, map_x, map_y, mask, map_batch, ho, wo, mode, static_cast<RawBf16*>(dst));
    else if (dtype == PF_F16)
        hipLaunchKernelGGL(k_remap<RawF16>, grid, block, 0, st, static_cast<const RawF16*>(src), B, C, hs,
                           ws, map_x, map_y, mask, map_batch, ho, wo, mode, static_cast<RawF16*>(dst));
    else
        PF_REQUIRE(false, "pf_remap: unsupported dtype %d", dtype);
    PF_CHECK_LAUNCH("pf_remap");
    return PF_OK;
}

extern "C" pf_status pf_equi_coords(int H, int W, float* lonlat, void* stream) {
    PF_REQUIRE(H > 1 && W > 1 && lonlat, "pf_equi_coords: bad arguments");
    hipLaunchKernelGGL(k_equi_coords, dim3(cdiv(static_cast<long>(H) * W, 256)), dim3(256), 0,
                       as_stream(stream), H, W, lonlat);
    PF_CHECK_LAUNCH("pf_equi_coords");
    return PF_OK;
}

extern "C" pf_status pf_spherical_pe(const float* coords, long n, const float* freq, int nfreq,
                                     float* out, void* stream) {
    PF_REQUIRE(coords && freq && out && n > 0 && nfreq > 0, "pf_spherical_pe: bad arguments");
    hipLaunchKernelGGL(k_spherical_pe, dim3(cdiv(n * nfreq, 256)), dim3(256), 0, as_stream(stream),
                       coords, n, freq, nfreq, out);
    PF_CHECK_LAUNCH("pf_spherical_pe");
    return PF_OK;
}

// ---- py360convert.e2p (dataset-side view cropping) ------------------------------------------------------
// external/py360convert/e2p.py:6-43 + utils.py:67-133,231-243: rays (x, -y, 1) with x / y float32 linspaces over
// +-tan(fov/2), rotated by Rx(v) Ry(u) Ri(in_rot) (row vector times matrix, double), lon = atan2(x, z),
// lat = atan2(y, sqrt(x^2 + z^2)), pixel = ((lon / 2pi + 0.5) W - 0.5, (-lat / pi + 0.5) H - 0.5), sampled by
// scipy.ndimage.map_coordinates(order 0 / 1, mode='wrap') on the image extended by two rows (last and first row
// rolled by W/2: the poles).  scipy's legacy 'wrap' has period len - 1 and computes in double; integer outputs
// are rounded half up and clamped -- restated literally (checked bit for bit against scipy in oracle/py360.py).
struct P360Cam { double Rx[9], Ry[9], Ri[9], x_max, y_max; };
constexpr int P360_BATCH = 16;
struct P360Batch { P360Cam c[P360_BATCH]; };

__device__ __forceinline__ double scipy_wrap_coord(double in, long len) {
    if (len <= 1) return 0.0;
    const double sz = static_cast<double>(len - 1);
    if (in < 0) in += sz * static_cast<double>(static_cast<long>(-in / sz) + 1);
    else if (in > sz) in -= sz * static_cast<double>(static_cast<long>(in / sz));
    return in;
}
__device__ __forceinline__ long scipy_wrap_index(long idx, long len) {
    const long s2 = len - 1;
    if (s2 <= 0) return 0;
    if (idx < 0) idx += s2 * (-idx / s2 + 1);
    else if (idx >= len) idx -= s2 * (idx / s2);
    return idx;
}

template <typename E> __device__ __forceinline__ E p360_store(double t);
template <> __device__ __forceinline__ float p360_store<float>(double t) { return static_cast<float>(t); }
template <> __device__ __forceinline__ uint8_t p360_store<uint8_t>(double t) {   // scipy CASE_INTERP_OUT_UINT
    t = t > 0 ? t + 0.5 : 0.0;
    t = t > 255.0 ? 255.0 : t;
    return static_cast<uint8_t>(t);
}

template <typename E>
__global__ void k_py360_e2p(const P360Batch cams, int ncam, const E* __restrict__ img, int H, int W, int C,
                            int oh, int ow, int order, E* __restrict__ out) {
    const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
    if (i >= static_cast<long>(ncam) * oh * ow) return;
    const int xo = i % ow, yo = (i / ow) % oh, cam = i / (static_cast<long>(ow) * oh);
    const P360Cam& cp = cams.c[cam];
    // np.linspace(-m, m, num, dtype=float32): float64 arithmetic, cast to float32
    const double vx = static_cast<double>(static_cast<float>(linspace_at(-cp.x_max, cp.x_max, ow, xo)));
    const double vy = -static_cast<double>(static_cast<float>(linspace_at(-cp.y_max, cp.y_max, oh, yo)));
    double v[3] = {vx, vy, 1.0}, t[3];
    const double* Rs[3] = {cp.Rx, cp.Ry, cp.Ri};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double* R = Rs[r];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = v[0] * R[k] + v[1] * R[3 + k] + v[2] * R[6 + k];
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2];
    }
    const double lon = atan2(v[0], v[2]);
    const double lat = atan2(v[1], sqrt(v[0] * v[0] + v[2] * v[2]));
    const double cx = (lon / (2 * kPi) + 0.5) * W - 0.5;
    const double cy = (-lat / kPi + 0.5) * H - 0.5;
    const long HP = H + 2;                                  // rows H, H+1: last / first row rolled by W/2
    const double y = scipy_wrap_coord(cy, HP), x = scipy_wrap_coord(cx, W);
    auto pixel = [&](long iy, long ix, int c) -> double {
        iy = scipy_wrap_index(iy, HP);
        ix = scipy_wrap_index(ix, W);
        if (iy >= H) {                                      // np.roll(row, W // 2): padded[j] = row[(j - W/2) mod W]
            long src = (ix - W / 2) % W;
            if (src < 0) src += W;
            ix = src;
            iy = iy == H ? H - 1 : 0;
        }
        return static_cast<double>(img[(iy * W + ix) * C + c]);
    };
    E* dst = out + i * C;
    if (order == 0) {
        const long iy = static_cast<long>(floor(y + 0.5)), ix = static_cast<long>(floor(x + 0.5));
        for (int c = 0; c < C; ++c) dst[c] = p360_store<E>(pixel(iy, ix, c));
    } else {
        const double fy0 = floor(y), fx0 = floor(x);
        const long y0 = static_cast<long>(fy0), x0 = static_cast<long>(fx0);
        const double wy1 = y - fy0, wx1 = x - fx0, wy0 = 1.0 - wy1, wx0 = 1.0 - wx1;
        for (int c = 0; c < C; ++c) {
            double acc = 0.0;
            acc = acc + pixel(y0, x0, c) * wy0 * wx0;
            acc = acc + pixel(y0, x0 + 1, c) * wy0 * wx1;
            acc = acc + pixel(y0 + 1, x0, c) * wy1 * wx0;
            acc = acc + pixel(y0 + 1, x0 + 1, c) * wy1 * wx1;
            dst[c] = p360_store<E>(acc);
        }
    }
}

// utils.py:231-243 rotation_matrix(rad, ax), same operation order, double
static void p360_rotation(double rad, const double ax_in[3], double R[9]) {
    double n = sqrt(ax_in[0] * ax_in[0] + ax_in[1] * ax_in[1] + ax_in[2] * ax_in[2]);
    double ax[3] = {ax_in[0] / n, ax_in[1] / n, ax_in[2] / n};
    const double c = cos(rad), s = sin(rad);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] = ((i == j) ? c : 0.0) + (ax[i] * ax[j]) * (1.0 - c);
    const double a[3] = {ax[0] * s, ax[1] * s, ax[2] * s};
    const double K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
    for (int i = 0; i < 9; ++i) R[i] = R[i] + K[i];
}

extern "C" pf_status pf_py360_e2p(const void* img, int dtype, int H, int W, int C, const double* host_hfov,
                                  const double* host_vfov, const double* host_u, const double* host_v,
                                  const double* host_in_rot, int ncam, int oh, int ow, int order, void* out,
                                  void* stream) {
    PF_REQUIRE(img && out && host_hfov && host_vfov && host_u && host_v && ncam > 0, "pf_py360_e2p: null pointer");
    PF_REQUIRE(H > 1 && W > 1 && C > 0 && oh > 0 && ow > 0, "pf_py360_e2p: bad sizes");
    PF_REQUIRE(order == 0 || order == 1, "pf_py360_e2p: order must be 0 (nearest) or 1 (bilinear)");
    PF_REQUIRE(dtype == PF_F32 || dtype == PF_U8, "pf_py360_e2p: dtype must be PF_F32 or PF_U8");
    hipStream_t st = as_stream(stream);
    const long per = static_cast<long>(oh) * ow;
    for (int c0 = 0; c0 < ncam; c0 += P360_BATCH) {
        const int n = ncam - c0 < P360_BATCH ? ncam - c0 : P360_BATCH;
        P360Batch b;
        for (int i = 0; i < n; ++i) {
            // e2p.py:16-32: degrees -> radians exactly as the reference spells it (x * np.pi / 180)
            const double hf = host_hfov[c0 + i] * kPi / 180, vf = host_vfov[c0 + i] * kPi / 180;
            const double u = -host_u[c0 + i] * kPi / 180, v = host_v[c0 + i] * kPi / 180;
            const double rot = (host_in_rot ? host_in_rot[c0 + i] : 0.0) * kPi / 180;
            P360Cam& cp = b.c[i];
            cp.x_max = tan(hf / 2);
            cp.y_max = tan(vf / 2);
            const double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0};
            p360_rotation(v, ex, cp.Rx);
            p360_rotation(u, ey, cp.Ry);
            double z1[3], z2[3];                             // np.array([0, 0, 1.0]).dot(Rx).dot(Ry)
            for (int k = 0; k < 3; ++k) z1[k] = 0.0 * cp.Rx[k] + 0.0 * cp.Rx[3 + k] + 1.0 * cp.Rx[6 + k];
            for (int k = 0; k < 3; ++k) z2[k] = z1[0] * cp.Ry[k] + z1[1] * cp.Ry[3 + k] + z1[2] * cp.Ry[6 + k];
            p360_rotation(rot, z2, cp.Ri);
        }
        for (int i = n; i < P360_BATCH; ++i) b.c[i] = b.c[0];
        const dim3 grid(cdiv(n * per, 256)), block(256);
        if (dtype == PF_F32)
            hipLaunchKernelGGL(k_py360_e2p<float>, grid, block, 0, st, b, n, static_cast<const float*>(img), H, W, C, oh, ow,
                               order, static_cast<float*>(out) + c0 * per * C);
        else
            hipLaunchKernelGGL(k_py360_e2p<uint8_t>, grid, block, 0, st, b, n, static_cast<const uint8_t*>(img), H, W, C, oh,
                               ow, order, static_cast<uint8_t*>(out) + c0 * per * C);
    }
    PF_CHECK_LAUNCH("pf_py360_e2p");
    return PF_OK;
}

static size_t round256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

extern "C" size_t pf_epa_tables_workspace_size(int ncam, int ph, int pw, int eh, int ew) {
    size_t dense = round256(static_cast<size_t>(ncam) * ph * pw * eh * ew * sizeof(float));
    return 3 * dense + round256(sizeof(CamParams) * static_cast<size_t>(ncam));
}

extern "C" pf_status pf_epa_tables_build(const double* fov, const double* theta, const double* phi,
                                         int m, int ph, int pw, int eh, int ew, float* bias_e,
                                         float* bias_p, uint8_t* flags_e, uint8_t* flags_p,
                                         void* workspace, size_t ws_bytes, void* stream) {
    PF_REQUIRE(fov && theta && phi && m > 0, "pf_epa_tables_build: cameras missing");
    PF_REQUIRE(ph > 1 && pw > 1 && eh > 1 && ew > 1, "pf_epa_tables_build: sizes must be > 1");
    PF_REQUIRE(bias_e && bias_p && flags_e && flags_p && workspace, "pf_epa_tables_build: null pointer");
    PF_REQUIRE(ws_bytes >= pf_epa_tables_workspace_size(m, ph, pw, eh, ew),
               "pf_epa_tables_build: workspace too small (%zu < %zu)", ws_bytes,
               pf_epa_tables_workspace_size(m, ph, pw, eh, ew));
    const long E = static_cast<long>(eh) * ew, P = static_cast<long>(ph) * pw, mP = m * P;
    const size_t smem_v = 2 * P * sizeof(float), smem_e = 2 * E * sizeof(float);
    PF_REQUIRE(smem_v <= 160 * 1024 && smem_e <= 160 * 1024,
               "pf_epa_tables_build: image too large for the LDS blur (%ld / %ld pixels)", P, E);
    hipStream_t st = as_stream(stream);
    const size_t dense = round256(static_cast<size_t>(m) * E * P * sizeof(float));
    char* ws = static_cast<char*>(workspace);
    float* W1 = reinterpret_cast<float*>(ws);
    float* W2 = reinterpret_cast<float*>(ws + dense);
    float* Bt = reinterpret_cast<float*>(ws + 2 * dense);
    CamParams* cams = reinterpret_cast<CamParams*>(ws + 3 * dense);
    for (int c0 = 0; c0 < m; c0 += CAM_BATCH) {            // camera constants into the caller's workspace, by value
        const int n = m - c0 < CAM_BATCH ? m - c0 : CAM_BATCH;
        CamBatch batch;
        fill_batch(fov, theta, phi, c0, n, ph, pw, &batch);
        constexpr int WORDS = sizeof(CamParams) / sizeof(double);
        hipLaunchKernelGGL(k_store_cams, dim3(cdiv(n * WORDS, 256)), dim3(256), 0, st, batch, n, cams + c0);
    }
    const int fe_ld = static_cast<int>(cdiv(mP, 32)), fp_ld = static_cast<int>(cdiv(E, 32));
    hipMemsetAsync(ws, 0, 3 * dense, st);
    hipMemsetAsync(bias_e, 0, sizeof(float) * E * mP, st);
    hipMemsetAsync(bias_p, 0, sizeof(float) * E * mP, st);
    hipMemsetAsync(flags_e, 0, cdiv(E, 32) * fe_ld, st);
    hipMemsetAsync(flags_p, 0, cdiv(mP, 32) * fp_ld, st);
    hipLaunchKernelGGL(k_scatter_p2e, dim3(cdiv(m * E, 256)), dim3(256), 0, st, cams, m, ph, pw, eh, ew, W1);
    hipLaunchKernelGGL(k_scatter_e2p, dim3(cdiv(m * P, 256)), dim3(256), 0, st, cams, m, ph, pw, eh, ew, W2);
    hipLaunchKernelGGL(k_crossfill, dim3(cdiv(m * E * P, 256)), dim3(256), 0, st, W1, W2, Bt, m, E, P);
    // kornia gaussian kernel (5 taps, sigma 1), normalised, fp32
    float g[5], gs = 0.0f;
    for (int k = 0; k < 5; ++k) { float x = static_cast<float>(k - 2); g[k] = expf(-(x * x) / 2.0f); gs += g[k]; }
    for (int k = 0; k < 5; ++k) g[k] /= gs;
    if (smem_v > 64 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_blur_normalise<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem_v));
    if (smem_e > 64 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_blur_normalise<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem_e));
    // view-side images: rows (v, e), ph x pw pixels  -> bias_e
    hipLaunchKernelGGL(k_blur_normalise<false>, dim3(m * E), dim3(256), smem_v, st, W2,
                       static_cast<int>(E), ph, pw, g[0], g[1], g[2], bias_e, E, mP, flags_e, fe_ld);
    // pano-side images: rows (v, p), eh x ew pixels -> bias_p
    hipLaunchKernelGGL(k_blur_normalise<true>, dim3(m * P), dim3(256), smem_e, st, Bt,
                       static_cast<int>(P), eh, ew, g[0], g[1], g[2], bias_p, E, mP, flags_p, fp_ld);
    PF_CHECK_LAUNCH("pf_epa_tables_build");
    return PF_OK;
}
