// Resolution-doubling implicit GEMMs in ONE launch: data gradient of the stride-2 3x3x3 conv and forward of the
// 2x2x2 stride-2 transposed conv.
//
// Both map a low-resolution tensor (C channels, contraction) to a tensor of twice the extent (M channels):
//     out[2q + par] = sum over the taps t of parity class par:  W[slot(par, t)] . in[q + off(par, t)],  off in {0,1}^3
//   * conv stride-2 dgrad  (dx from dy):   par = 0 -> tap d = 1 (off 0);  par = 1 -> d = 0 (off +1), d = 2 (off 0)
//                                          per dimension, i.e. 1/2/2/4/2/4/4/8 = 27 taps over the 8 classes;
//   * transposed conv k2s2 (y from x):     one tap per class, slot = class, off 0.
// The first version launched the generic kernel once per class (8 launches, each re-reading the input and writing
// every other voxel of the output: 64-byte pieces with 64-byte holes).  Here a wave owns 32 low-res voxels x 32
// output channels and keeps the accumulators of ALL EIGHT classes (8 x 16 registers); the low-res halo tile is
// staged once per 16-channel chunk and feeds all 27 (8) MFMAs, and the eight classes of a voxel leave the CU
// together, so every output line is written once and completely.  These layers are bound by the output write
// (4x the input for the first decoder / encoder level), not by the matrix pipe.
//   * 256-thread blocks (4 waves: 2 z-planes x 2 y-halves of a 2x8x8 low-res tile), two blocks per CU;
//   * double-buffered LDS: halo tile 3x9x9 positions x 32 B (7.6 KB), weights NTAP x 32 rows x 32 B (27 / 8 KB);
//     next chunk's global loads are issued before the MFMAs and written to the other buffers after them: one
//     barrier per chunk;
//   * same conflict-free LDS layout as the stride-1 kernel (igemm_conv_tile.hip): 32-byte rows, 16-byte half XOR-keyed
//     with the halo row parity (B) / bit 3 of the output channel (A).
#include "igemm_common.h"

namespace {

constexpr int TZ = 2, TY = 8, TX = 8, PZ = 3, PY = 9, PX = 9;
constexpr int P = PZ * PY * PX;              // 243 halo positions
constexpr int CK = 16, ROWB = 32, MB = 32;
constexpr int XBYTES = P * ROWB;             // 7776
constexpr int XCHUNKS = P * 2;
constexpr int NT = 256;
constexpr int XN = (XCHUNKS + NT - 1) / NT;  // 2

template <int MODE>
struct UpCfg {
    static constexpr int NTAP = MODE == 0 ? 27 : 8;
    static constexpr int WBYTES = NTAP * MB * ROWB;
    static constexpr int WCHUNKS = NTAP * MB * 2;
    static constexpr int WN = (WCHUNKS + NT - 1) / NT;
};

__device__ __forceinline__ int up_xaddr(int pz, int py, int px, int c2) {
    return ((pz * PY + py) * PX + px) * ROWB + ((c2 ^ (py & 1)) << 4);
}
__device__ __forceinline__ int up_waddr(int row, int c2) { return row * ROWB + ((c2 ^ ((row >> 3) & 1)) << 4); }

// same lane -> (row, x) map as the stride-1 kernel: every 16-lane ds_read_b128 group covers two full 8-voxel rows
__device__ __forceinline__ void up_lane_voxel(int v, int& r, int& x) {
    if (v < 4) { r = 0; x = v; }
    else if (v < 12) { r = 2; x = v - 4; }
    else if (v < 16) { r = 0; x = v - 8; }
    else if (v < 20) { r = 3; x = v - 16; }
    else if (v < 28) { r = 1; x = v - 20; }
    else { r = 3; x = v - 24; }
}

// MODE 0: stride-2 conv dgrad (27 taps, slot = dz*9 + dy*3 + dx of the dgrad panel); MODE 1: convT k2s2 fwd
template <int MODE>
__global__ __launch_bounds__(NT, 2) void igemm_up2_kernel(const ConvParams p, int mblocks) {
    using Cfg = UpCfg<MODE>;
    constexpr int WN = Cfg::WN, WBYTES = Cfg::WBYTES, WCHUNKS = Cfg::WCHUNKS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xb = smem;
    char* const wb = smem + 2 * XBYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v = lane & 31, hk = lane >> 5;
    const int m0 = (blockIdx.x % mblocks) * MB;
    int t = blockIdx.x / mblocks;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; t /= p.tiles_y;
    const int tz = t % p.tiles_z; t /= p.tiles_z;
    const int n = t, qz0 = tz * TZ, qy0 = ty * TY, qx0 = tx * TX;
    const int nchunks = (p.C + CK - 1) / CK;
    const bool interior = qz0 + PZ <= p.Di && qy0 + PY <= p.Hi && qx0 + PX <= p.Wi && (p.C % CK) == 0;

    int xrel[XN], xlds[XN];
#pragma unroll
    for (int i = 0; i < XN; ++i) {
        const int idx = min(i * NT + tid, XCHUNKS - 1);
        const int pos = idx >> 1, c2 = idx & 1;
        const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
        xrel[i] = ((pz * p.Hi + py) * p.Wi + px) * p.ld_x + c2 * 8;
        xlds[i] = up_xaddr(pz, py, px, c2);
    }
    int wrel[WN], wlds[WN];
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        const int idx = min(i * NT + tid, WCHUNKS - 1);
        const int c2 = idx & 1, row = idx >> 1, r = row % MB, tl = row / MB;
        wrel[i] = (tl * MB + r) * 16 + c2 * 8;     // == idx * 8: the blocked panel chunk is one contiguous run
        wlds[i] = up_waddr(row, c2);
    }

    half8 xr[XN], wr[WN];
    unsigned xok = 0;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const long xbase = ((((long)n * p.Di + qz0) * p.Hi + qy0) * p.Wi + qx0) * p.ld_x;
    const int nck16 = p.KCpad >> 4;
    const half_t* const wbase = p.wp + (long)(m0 >> 5) * nck16 * Cfg::NTAP * 512;

    // unconditional loads + masks (a predicated load serialises the prefetch, see igemm_conv_tile.hip)
    auto load_x = [&](int c0) {
        if (interior) {
            const half_t* bp = p.x + xbase + c0;
#pragma unroll
            for (int i = 0; i < XN; ++i) xr[i] = *reinterpret_cast<const half8*>(bp + xrel[i]);
            xok = 0xFFFFu;
        } else {
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < XN; ++i) {
                const int idx = min(i * NT + tid, XCHUNKS - 1);
                const int pos = idx >> 1, c2 = idx & 1;
                const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
                const bool ok = qz0 + pz < p.Di && qy0 + py < p.Hi && qx0 + px < p.Wi && c0 + c2 * 8 < p.C;
                xr[i] = *reinterpret_cast<const half8*>(p.x + (ok ? xbase + c0 + xrel[i] : 0));
                m |= (ok ? 1u : 0u) << i;
            }
            xok = m;
        }
    };
    auto store_x = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < XN; ++i)
            if (i * NT + tid < XCHUNKS) *reinterpret_cast<half8*>(buf + xlds[i]) = ((xok >> i) & 1u) ? xr[i] : zero8;
    };
    auto load_w = [&](int c0) {
#pragma unroll
        for (int i = 0; i < WN; ++i) wr[i] = *reinterpret_cast<const half8*>(wbase + (long)(c0 >> 4) * Cfg::NTAP * 512 + wrel[i]);
    };
    auto store_w = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < WN; ++i)
            if (i * NT + tid < WCHUNKS) *reinterpret_cast<half8*>(buf + wlds[i]) = wr[i];
    };

    // wave -> (z plane, y half) of the tile; lane -> (row, x)
    int vr, vx;
    up_lane_voxel(v, vr, vx);
    const int wz = wave >> 1, wy = (wave & 1) * 4 + vr;
    int lterm[2];    // byte address of halo position (wz, wy, vx); 16-byte half keyed with the parity of row wy + par
#pragma unroll
    for (int par = 0; par < 2; ++par) lterm[par] = ((wz * PY + wy) * PX + vx) * ROWB + ((hk ^ ((wy + par) & 1)) << 4);
    const int a_lane = v * ROWB + ((hk ^ ((v >> 3) & 1)) << 4);

    floatx16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

    load_x(0);
    load_w(0);
    store_x(xb);
    store_w(wb);
    __syncthreads();

#pragma unroll 1
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        const char* xl = xb + (ch & 1) * XBYTES;
        const char* wl = wb + (ch & 1) * WBYTES;
        if (more) {
            load_x((ch + 1) * CK);
            load_w((ch + 1) * CK);
        }
        if (MODE == 0) {
            half8 fb[2][2][2];
#pragma unroll
            for (int oz = 0; oz < 2; ++oz)
#pragma unroll
                for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                    for (int ox = 0; ox < 2; ++ox)
                        fb[oz][oy][ox] = *reinterpret_cast<const half8*>(xl + ((oz * PY + oy) * PX + ox) * ROWB + lterm[oy]);
#pragma unroll
            for (int cls = 0; cls < 8; ++cls) {
                const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
#pragma unroll
                for (int a = 0; a <= pz; ++a)
#pragma unroll
                    for (int b = 0; b <= py; ++b)
#pragma unroll
                        for (int c = 0; c <= px; ++c) {
                            const int dz = pz ? 2 * a : 1, dy = py ? 2 * b : 1, dx = px ? 2 * c : 1;
                            const int oz = pz ? 1 - a : 0, oy = py ? 1 - b : 0, ox = px ? 1 - c : 0;
                            const int slot = dz * 9 + dy * 3 + dx;
                            const half8 fa = *reinterpret_cast<const half8*>(wl + slot * MB * ROWB + a_lane);
                            acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb[oz][oy][ox], acc[cls], 0, 0, 0);
                        }
            }
        } else {
            const half8 fb = *reinterpret_cast<const half8*>(xl + lterm[0]);
#pragma unroll
            for (int cls = 0; cls < 8; ++cls) {
                const half8 fa = *reinterpret_cast<const half8*>(wl + cls * MB * ROWB + a_lane);
                acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[cls], 0, 0, 0);
            }
        }
        if (more) {
            store_x(xb + ((ch + 1) & 1) * XBYTES);
            store_w(wb + ((ch + 1) & 1) * WBYTES);
        }
        __syncthreads();
    }

    // ---- epilogue: through LDS, so that every global store instruction writes whole output ROWS -------------
    // A lane's accumulators are 4-channel (8-byte) pieces of eight output voxels that lie two apart: stored directly, a
    // wave-wide store touched 32 cache lines with 16 bytes each and the kernel was bound by the address pipe (2.4-2.9 TB/s
    // of output).  The block's output tile (4 x 16 x 16 voxels x 32 channels) is staged per z parity (2 x 16 rows of
    // 16 voxels x 64 B) and leaves as 16-byte-per-lane stores of complete 1 KB rows (8 full lines per instruction).
    // Staging layout: row stride 1040 B, the 8-byte piece index p = 2 qq + hk XOR-keyed with (vx >> 1) & 3: the
    // 32 lanes of a ds_write_b64 half hit 32 distinct bank pairs, the ds_read_b128 of 4 consecutive voxels is linear.
    constexpr int RS = 1040;
    char* const stg = smem;
    const int qz = qz0 + wz, qy = qy0 + wy;
    const int skey = (vx >> 1) & 3;
    // accumulate mode (the encoder's strided dgrad adds to the skip gradient the decoder wrote): the 8 old rows of a pass are
    // loaded BEFORE its staging writes and barrier, all in flight together -- inside the store loop each read-modify-write was a
    // dependent global round trip (16 per block; the kernel ran at 0.12 busy matrix pipes and 3.2 TB/s)
    const int eox = (lane >> 2) & 15, ec = lane & 3;
    const int egx = 2 * qx0 + eox, em = m0 + ec * 8;
    auto out_ptr = [&](int pz, int rr) -> half8* {
        const int row = wave * 8 + rr, zl = row >> 4, oyl = row & 15;
        const int oz = 2 * (qz0 + zl) + pz, oy = 2 * qy0 + oyl;
        const bool ok = oz < p.Do && oy < p.Ho && egx < p.Wo && em < p.M;
        return ok ? reinterpret_cast<half8*>(p.y + ((((long)n * p.Do + oz) * p.Ho + oy) * p.Wo + egx) * p.ld_y + em) : nullptr;
    };
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
        half8 olds[8];
        if (p.accumulate) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const half8* d = out_ptr(pz, rr);
                olds[rr] = *(d ? d : reinterpret_cast<const half8*>(p.y));
            }
        }
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int cls = pz * 4 + c4, py = c4 >> 1, px = c4 & 1;
            char* vb = stg + (wz * 16 + 2 * wy + py) * RS + (2 * vx + px) * 64;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const half4 o = {(half_t)acc[cls][qq * 4 + 0], (half_t)acc[cls][qq * 4 + 1], (half_t)acc[cls][qq * 4 + 2],
                                 (half_t)acc[cls][qq * 4 + 3]};
                *reinterpret_cast<half4*>(vb + (((2 * qq + hk) ^ skey) << 3)) = o;
            }
        }
        __syncthreads();
        {
            const int rkey = (eox >> 2) & 3;                             // = skey of the low-res voxel ox >> 1
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int row = wave * 8 + rr;
                half8 v = *reinterpret_cast<const half8*>(stg + row * RS + eox * 64 + ((ec ^ (rkey >> 1)) << 4));
                if (rkey & 1) v = half8{v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
                half8* dst = out_ptr(pz, rr);
                if (dst) {
                    if (p.accumulate) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)olds[rr][e]);
                    }
                    *dst = v;
                }
            }
        }
        if (pz == 0) __syncthreads();
    }
    (void)qz; (void)qy;
}

template <int MODE>
int launch_up2(hipStream_t s, ConvParams& p, const char* name) {
    using Cfg = UpCfg<MODE>;
    p.tiles_z = lnn_cdiv((p.Do + 1) / 2, TZ); p.tiles_y = lnn_cdiv((p.Ho + 1) / 2, TY); p.tiles_x = lnn_cdiv((p.Wo + 1) / 2, TX);
    const int mblocks = lnn_cdiv(p.M, MB);
    const long blocks = (long)p.N * p.tiles_z * p.tiles_y * p.tiles_x * mblocks;
    const size_t lds_main = 2 * XBYTES + 2 * Cfg::WBYTES, lds_stage = 32 * 1040;
    const size_t lds = lds_main > lds_stage ? lds_main : lds_stage;
    auto kern = igemm_up2_kernel<MODE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), lds, s, p, mblocks);
    LNN_CHECK_LAUNCH(name);
    return LNN_OK;
}

}  // namespace

// p.x: low-res input (C channels, extents Di/Hi/Wi), p.y: output (M channels, extents Do/Ho/Wo <= 2 * input)
int lnn_launch_up2_dgrad(hipStream_t s, ConvParams& p, const char* name) { return launch_up2<0>(s, p, name); }
int lnn_launch_up2_convT(hipStream_t s, ConvParams& p, const char* name) { return launch_up2<1>(s, p, name); }
