// MFCC + delta + delta-delta + z-score extraction on gfx950.
//
// Two kernels:
//   mfcc_base_kernel  PCM -> 20 un-normalised MFCCs per window (reference lib.rs:292-315).
//                     A wavefront works on 6 windows at a time, 10 lanes per window; the
//                     800-point real DFT is a 400-point complex FFT done as two register
//                     passes of 20-point DFTs with one exchange through LDS (see
//                     extract_phases.h).  The tables the phases read are copied into LDS once per
//                     workgroup; after that the waves never synchronise with each other.
//   finish_kernel     deltas with edge replication at file boundaries, concat and per-frame
//                     z-score (lib.rs:212-228, 321-342); streams 80 B in, 240 B out per window.
#include <algorithm>
#include <cstring>
#include <new>

#include "extract_phases.h"
#include "internal.h"
#include "prof.h"
#include "rng.h"
#include "streamz_hip.h"

namespace sz {

namespace {

// Lanes of one wavefront exchange data through LDS between phases.  LDS operations of a
// wavefront execute in program order, so only the compiler has to be stopped from moving
// accesses across the phase boundary.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kBaseThreads = 256;
constexpr int kBaseWaves = kBaseThreads / 64;
constexpr size_t kRegionBytes = (size_t)kBaseWaves * kWinPerWave * kRegionStride * sizeof(float);
// the windows' scratch regions, then the workgroup's copy of the tables
constexpr size_t kBaseLdsBytes = kRegionBytes + sizeof(KernelTables);
static_assert(kRegionBytes % 16 == 0 && 3 * kBaseLdsBytes <= 160 * 1024, "three workgroups per CU");

__global__ __launch_bounds__(kBaseThreads, 3) void mfcc_base_kernel(const int16_t* __restrict__ pcm,
                                                                  const Segment* __restrict__ segs,
                                                                  uint32_t n_segs,
                                                                  const KernelTables* __restrict__ Tg,
                                                                  float* __restrict__ base,
                                                                  const uint64_t* __restrict__ shift_p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // samples every unit of this launch lies further into the PCM buffer than its table says (a plan
    // that walks along a recording; the finish kernel of the previous launch moved it on)
    const uint64_t shift = shift_p ? *shift_p : 0;
    // the tables into LDS, once per workgroup (the only workgroup barrier of the kernel)
    const KernelTables* T = reinterpret_cast<const KernelTables*>(reinterpret_cast<const char*>(lds) + kRegionBytes);
    {
        const float4* src = reinterpret_cast<const float4*>(Tg);
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<char*>(lds) + kRegionBytes);
        for (int i = threadIdx.x; i < (int)(sizeof(KernelTables) / 16); i += kBaseThreads) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // lanes 60..63 shadow lanes 50..53: same inputs, same LDS writes, no global store
    const int slot = lane < 60 ? lane : lane - 10;
    const int grp = slot / kGroup;
    const int u = slot - grp * kGroup;
    float* R = lds + (wave * kWinPerWave + grp) * kRegionStride;
    const uint32_t waves_total = gridDim.x * kBaseWaves;
    for (uint32_t sg = blockIdx.x * kBaseWaves + wave; sg < n_segs; sg += waves_total) {
        const Segment S = segs[sg];
        for (uint32_t w0 = 0; w0 < S.n_win; w0 += kWinPerWave) {
            uint32_t wi = w0 + grp;
            const bool valid = wi < S.n_win;
            if (!valid) wi = S.n_win - 1;
            const int16_t* win = pcm + S.pcm_off + shift + (uint64_t)wi * kHop;
            // The tables are re-read from LDS in every iteration: letting the compiler hoist ~250
            // loop-invariant twiddle loads into registers spills.  (The opaque value is an offset,
            // not the pointer, so the address space of the loads stays known.)
            uint32_t opaque0 = 0;
            asm volatile("" : "+s"(opaque0));
            {
                v2f x0[20];
                {
                    v2f hi_a[kHalfRows], hi_b[kHalfRows];
                    phase1(u, win, R, T + opaque0, hi_a, hi_b);
                    wave_sync();
                    read_row(R, first_row(u), x0);
                    wave_sync();
                    phase1_high(u, R, hi_a, hi_b);
                }
                wave_sync();
                asm volatile("" : "+s"(opaque0));
                phase2(u, R, T + opaque0, x0);
            }
            wave_sync();
            {
                float p[kMelSteps];
                phase3_load(u, R, p);
                wave_sync();
                asm volatile("" : "+s"(opaque0));
                phase3_sweep(u, R, T + opaque0, p);
            }
            wave_sync();
            phase4(u, R, T + opaque0);
            wave_sync();
            asm volatile("" : "+s"(opaque0));
            float c_lo, c_hi;
            phase5(u, R, T + opaque0, c_lo, c_hi);
            if (valid && lane < 60) {
                float* o = base + (S.win_first + wi) * kMfcc;
                o[u] = c_lo;
                o[u + 10] = c_hi;
            }
            wave_sync();
        }
    }
}

// augment, lib.rs:103-116: circular shift, gain, additive noise, clamp, truncating cast.  The
// three roundings (two products, one sum) are kept separate, as the reference computes them.
__global__ void augment_kernel(const int16_t* __restrict__ in, uint64_t n, float gain, float level, uint64_t shift,
                               const float* __restrict__ noise, uint64_t seed, uint64_t counter0, int16_t* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t idx = i + shift;
    if (idx >= n) idx -= n;
    const float nz = noise ? noise[i] : (2.0f * uniform01(seed, counter0 + i, 0xA065ull) - 1.0f) * level;
    float v = __fadd_rn(__fmul_rn((float)in[idx], gain), __fmul_rn(nz, 32767.0f));
    v = fminf(fmaxf(v, -32768.0f), 32767.0f);
    out[i] = (int16_t)(int)v;
}

constexpr int kFinishThreads = kSegWindows * 4;

__device__ __forceinline__ float quad_sum(float v)
{
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    return v;
}

__global__ __launch_bounds__(kFinishThreads) void finish_kernel(const float* __restrict__ base,
                                                               const Segment* __restrict__ segs,
                                                               uint32_t n_segs, float* __restrict__ out,
                                                               uint64_t* __restrict__ shift_p, uint64_t advance)
{
    __shared__ __attribute__((aligned(16))) float F[kSegWindows * kFeat];
    // the plan moves on for its next launch (this kernel reads no PCM, and the next MFCC launch starts
    // after this one has ended)
    if (shift_p && blockIdx.x == 0 && threadIdx.x == 0) *shift_p += advance;
    const int f = threadIdx.x >> 2;
    const int p = threadIdx.x & 3;
    for (uint32_t sg = blockIdx.x; sg < n_segs; sg += gridDim.x) {
        const Segment S = segs[sg];
        const bool active = (uint32_t)f < S.n_win;
        const int64_t n = (int64_t)S.file_n_win;
        const int64_t t = active ? (int64_t)(S.win_first - S.file_win_first) + f : 0;
        const float* fb = base + S.file_win_first * kMfcc;
        // rows t-2 .. t+2 with the edge rule of add_deltas (lib.rs:215-220) applied twice
        const int64_t tm1 = t > 0 ? t - 1 : t;
        const int64_t tp1 = t + 1 < n ? t + 1 : t;
        const int64_t tm1m = tm1 > 0 ? tm1 - 1 : tm1;
        const int64_t tm1p = tm1 + 1 < n ? tm1 + 1 : tm1;
        const int64_t tp1m = tp1 > 0 ? tp1 - 1 : tp1;
        const int64_t tp1p = tp1 + 1 < n ? tp1 + 1 : tp1;
        float c[5], d[5], dd[5];
        float sum = 0.0f;
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
            const int j = 5 * p + jj;
            c[jj] = fb[t * kMfcc + j];
            d[jj] = (fb[tp1 * kMfcc + j] - fb[tm1 * kMfcc + j]) / 2.0f;
            const float d_next = (fb[tp1p * kMfcc + j] - fb[tp1m * kMfcc + j]) / 2.0f;
            const float d_prev = (fb[tm1p * kMfcc + j] - fb[tm1m * kMfcc + j]) / 2.0f;
            dd[jj] = (d_next - d_prev) / 2.0f;
            sum += c[jj] + d[jj] + dd[jj];
        }
        const float mean = quad_sum(sum) / (float)kFeat;
        float vs = 0.0f;
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
            const float a = c[jj] - mean, b = d[jj] - mean, e = dd[jj] - mean;
            vs += a * a + b * b + e * e;
        }
        const float var = quad_sum(vs) / (float)kFeat;
        const float sd = fmaxf(sqrtf(var), 1e-6f);
        if (active) {
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) {
                const int j = 5 * p + jj;
                F[f * kFeat + j] = (c[jj] - mean) / sd;
                F[f * kFeat + kMfcc + j] = (d[jj] - mean) / sd;
                F[f * kFeat + 2 * kMfcc + j] = (dd[jj] - mean) / sd;
            }
        }
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(out + S.out_first * kFeat);
        const float4* src = reinterpret_cast<const float4*>(F);
        const int n4 = (int)S.n_win * (kFeat / 4);
        for (int i = threadIdx.x; i < n4; i += kFinishThreads) dst[i] = src[i];
        __syncthreads();
    }
}

}  // namespace

int extractor_create(Extractor** out)
{
    Extractor* e = new (std::nothrow) Extractor();
    if (!e) { set_error("out of host memory"); return SZH_ENOMEM; }
    const int rc = build_extract_tables(&e->host);
    if (rc != 0) {
        set_error("mel filterbank does not have the two-triangles-per-bin structure (code %d)", rc);
        delete e;
        return SZH_EINVAL;
    }
    hipError_t he = hipGetDevice(&e->device);
    KernelTables* kt = new (std::nothrow) KernelTables();
    if (!kt) { delete e; set_error("out of host memory"); return SZH_ENOMEM; }
    std::memcpy(kt->twa, e->host.twa, sizeof(kt->twa));
    std::memcpy(kt->twb, e->host.twb, sizeof(kt->twb));
    std::memcpy(kt->sw, e->host.sw, sizeof(kt->sw));
    std::memcpy(kt->pk, e->host.pk, sizeof(kt->pk));
    std::memcpy(kt->melw, e->host.melw, sizeof(kt->melw));
    std::memcpy(kt->mel_flush, e->host.mel_flush, sizeof(kt->mel_flush));
    std::memcpy(kt->comb_pos, e->host.comb_pos, sizeof(kt->comb_pos));
    std::memcpy(kt->dct2, e->host.dct2, sizeof(kt->dct2));
    if (he == hipSuccess) he = hipMalloc((void**)&e->dev, sizeof(KernelTables));
    if (he == hipSuccess) he = hipMemcpy(e->dev, kt, sizeof(KernelTables), hipMemcpyHostToDevice);
    delete kt;
    if (he != hipSuccess) {
        if (e->dev) (void)hipFree(e->dev);
        delete e;
        return hip_fail(he, "extractor tables upload");
    }
    *out = e;
    return SZH_OK;
}

void extractor_destroy(Extractor* e)
{
    if (!e) return;
    if (e->dev) (void)hipFree(e->dev);
    delete e;
}

// Windows per unit of work of the MFCC kernel (one wave walks a unit 6 windows at a time):
// small jobs get short units so that every SIMD has a wave, large jobs the full segment.
static uint32_t base_unit(uint64_t windows)
{
    const uint64_t u = (windows / 8192 + kWinPerWave - 1) / kWinPerWave * kWinPerWave;
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(u, kWinPerWave), kSegWindows);
}

int plan_create(Extractor* e, const uint64_t* offsets, const uint64_t* lengths, uint32_t n_files, Plan** out)
{
    Plan* p = new (std::nothrow) Plan();
    if (!p) { set_error("out of host memory"); return SZH_ENOMEM; }
    p->ext = e;
    p->win_off.assign((size_t)n_files + 1, 0);
    for (uint32_t f = 0; f < n_files; ++f) {
        if (offsets[f] % 8 != 0) {
            set_error("file %u: offset %llu is not a multiple of 8 samples", f, (unsigned long long)offsets[f]);
            delete p;
            return SZH_EINVAL;
        }
        const uint64_t nw = szh_window_count(lengths[f]);
        p->win_off[f + 1] = p->win_off[f] + nw;
        if (offsets[f] + lengths[f] > p->pcm_extent) p->pcm_extent = offsets[f] + lengths[f];
        if (nw > 0xffffffffull) {
            set_error("file %u has more than 2^32 windows", f);
            delete p;
            return SZH_EINVAL;
        }
    }
    p->total_windows = p->win_off[n_files];
    const uint32_t unit = base_unit(p->total_windows);
    std::vector<Segment> segs;
    for (int pass = 0; pass < 2; ++pass) {  // MFCC units, then the 48-window segments of the finish kernel
        const uint32_t step = pass == 0 ? unit : kSegWindows;
        if (pass == 1 && unit == kSegWindows) break;  // both kernels walk the same table
        for (uint32_t f = 0; f < n_files; ++f) {
            const uint64_t nw = p->win_off[f + 1] - p->win_off[f];
            for (uint64_t w = 0; w < nw; w += step) {
                Segment s{};
                s.pcm_off = offsets[f] + w * kHop;
                s.win_first = p->win_off[f] + w;
                s.out_first = s.win_first;
                s.file_win_first = p->win_off[f];
                s.n_win = (uint32_t)(nw - w < step ? nw - w : step);
                s.file_n_win = (uint32_t)nw;
                segs.push_back(s);
            }
        }
        if (pass == 0) p->n_segs = (uint32_t)segs.size();
        if (segs.size() > 0xffffffffull) { set_error("too many segments"); delete p; return SZH_EINVAL; }
    }
    p->n_fin = unit == kSegWindows ? p->n_segs : (uint32_t)segs.size() - p->n_segs;
    if (p->n_segs) {
        hipError_t he = hipMalloc((void**)&p->d_segs, segs.size() * sizeof(Segment));
        if (he == hipSuccess)
            he = hipMemcpy(p->d_segs, segs.data(), segs.size() * sizeof(Segment), hipMemcpyHostToDevice);
        if (he == hipSuccess) he = hipMalloc((void**)&p->d_base, p->total_windows * kMfcc * sizeof(float));
        if (he != hipSuccess) {
            plan_destroy(p);
            return hip_fail(he, "plan allocation");
        }
        p->d_fin = unit == kSegWindows ? p->d_segs : p->d_segs + p->n_segs;
    }
    *out = p;
    return SZH_OK;
}

int plan_create_tile(Extractor* e, uint64_t pcm_off, uint32_t lead, uint32_t n_out, uint32_t trail, Plan** out)
{
    if (pcm_off % 8 != 0) { set_error("tile offset %llu is not a multiple of 8 samples", (unsigned long long)pcm_off); return SZH_EINVAL; }
    Plan* p = new (std::nothrow) Plan();
    if (!p) { set_error("out of host memory"); return SZH_ENOMEM; }
    p->ext = e;
    const uint32_t nb = lead + n_out + trail;
    p->win_off = {0, n_out};
    p->total_windows = n_out;
    p->pcm_extent = nb ? pcm_off + (uint64_t)(nb - 1) * kHop + kWindow : 0;
    std::vector<Segment> segs;
    const uint32_t unit = base_unit(nb);
    for (uint32_t w = 0; w < nb; w += unit) {
        Segment s{};
        s.pcm_off = pcm_off + (uint64_t)w * kHop;
        s.win_first = w;
        s.n_win = nb - w < unit ? nb - w : unit;
        s.file_n_win = nb;
        segs.push_back(s);
    }
    p->n_segs = (uint32_t)segs.size();
    for (uint32_t i = 0; i < n_out; i += kSegWindows) {
        Segment s{};
        s.win_first = lead + i;
        s.out_first = i;
        s.n_win = n_out - i < kSegWindows ? n_out - i : kSegWindows;
        s.file_n_win = nb;
        segs.push_back(s);
    }
    p->n_fin = (uint32_t)segs.size() - p->n_segs;
    if (p->n_fin) {
        hipError_t he = hipMalloc((void**)&p->d_segs, segs.size() * sizeof(Segment));
        if (he == hipSuccess)
            he = hipMemcpy(p->d_segs, segs.data(), segs.size() * sizeof(Segment), hipMemcpyHostToDevice);
        if (he == hipSuccess) he = hipMalloc((void**)&p->d_base, (uint64_t)nb * kMfcc * sizeof(float));
        if (he != hipSuccess) {
            plan_destroy(p);
            return hip_fail(he, "tile plan allocation");
        }
        p->d_fin = p->d_segs + p->n_segs;
    } else {
        p->n_segs = 0;
    }
    *out = p;
    return SZH_OK;
}

int plan_set_advance(Plan* p, uint64_t samples)
{
    if (p->n_segs == 0) return SZH_OK;
    if (!p->d_shift) {
        SZ_HIP(hipMalloc((void**)&p->d_shift, sizeof(uint64_t)));
        SZ_HIP(hipMemset(p->d_shift, 0, sizeof(uint64_t)));
    }
    p->advance = samples;
    return SZH_OK;
}

void plan_destroy(Plan* p)
{
    if (!p) return;
    if (p->d_segs) (void)hipFree(p->d_segs);
    if (p->d_base) (void)hipFree(p->d_base);
    if (p->d_shift) (void)hipFree(p->d_shift);
    delete p;
}

int plan_run(Plan* p, const int16_t* pcm_dev, float* feats_dev, hipStream_t stream)
{
    if (p->n_segs == 0) return SZH_OK;
    if (((uintptr_t)pcm_dev & 15) != 0 || ((uintptr_t)feats_dev & 15) != 0) {
        set_error("device buffers must be 16-byte aligned");
        return SZH_EINVAL;
    }
    if (int rc = allow_dynamic_lds((const void*)mfcc_base_kernel, kBaseLdsBytes)) return rc;
    const uint32_t wave_jobs = (p->n_segs + kBaseWaves - 1) / kBaseWaves;
    const uint32_t grid_a = wave_jobs < 256u * 3u * 8u ? wave_jobs : 256u * 3u * 8u;
    {
        ProfScope prof("mfcc_base", stream);
        hipLaunchKernelGGL(mfcc_base_kernel, dim3(grid_a), dim3(kBaseThreads), kBaseLdsBytes, stream, pcm_dev,
                           p->d_segs, p->n_segs, p->ext->dev, p->d_base, p->d_shift);
    }
    SZ_HIP(hipGetLastError());
    const uint32_t grid_b = p->n_fin < 256u * 32u ? p->n_fin : 256u * 32u;
    {
        ProfScope prof("finish", stream);
        hipLaunchKernelGGL(finish_kernel, dim3(grid_b), dim3(kFinishThreads), 0, stream, p->d_base, p->d_fin,
                           p->n_fin, feats_dev, p->d_shift, p->advance);
    }
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

// ---- one clip without any allocation ------------------------------------------------------
// szh_extract_dev_noalloc runs the same two kernels as plan_run over one clip, with the two
// segment tables and the un-normalised MFCCs in a fixed device array instead of plan-owned
// allocations.  The tables are filled on the device (a plan uploads them from the host).
namespace {

constexpr uint32_t kNoAllocUnit = kWinPerWave;  // base_unit() of any job this small
constexpr uint32_t kNoAllocMaxUnits = (SZH_NOALLOC_MAX_WINDOWS + kNoAllocUnit - 1) / kNoAllocUnit;
constexpr uint32_t kNoAllocMaxFin = (SZH_NOALLOC_MAX_WINDOWS + kSegWindows - 1) / kSegWindows;
struct NoAllocScratch {
    Segment units[kNoAllocMaxUnits];                  // work units of mfcc_base_kernel
    Segment fin[kNoAllocMaxFin];                      // 48-window segments of finish_kernel
    float base[(size_t)SZH_NOALLOC_MAX_WINDOWS * kMfcc];  // [windows][20]
};
// Every GPU the library runs on holds its own copy of this array (1.4 MB) until the process ends.
// Host code never uses the name as an address: the per-device address comes from
// hipGetSymbolAddress (noalloc_scratch) and is what the kernels receive.
static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
NoAllocScratch* g_scratch_ptr[kMaxDevices] = {};

// the current device's copy of `scratch`; looked up once per device
int noalloc_scratch(NoAllocScratch** out)
{
    int dev = 0;
    SZ_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) { set_error("device ordinal %d is beyond the %d this call supports", dev, kMaxDevices); return SZH_EINVAL; }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        SZ_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (int rc = allow_dynamic_lds((const void*)mfcc_base_kernel, kBaseLdsBytes)) return rc;
        g_scratch_ptr[dev] = static_cast<NoAllocScratch*>(p);
    }
    *out = g_scratch_ptr[dev];
    return SZH_OK;
}

// the tables plan_create builds for a single file at offset 0 with n_win windows
__global__ void single_clip_segments_kernel(Segment* __restrict__ units, uint32_t n_units, Segment* __restrict__ fin,
                                            uint32_t n_fin, uint32_t n_win)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t step = pass == 0 ? kNoAllocUnit : kSegWindows;
        if (i >= (pass == 0 ? n_units : n_fin)) continue;
        const uint32_t w = i * step;
        Segment s{};
        s.pcm_off = (uint64_t)w * kHop;
        s.win_first = w;
        s.out_first = w;
        s.file_win_first = 0;
        s.n_win = n_win - w < step ? n_win - w : step;
        s.file_n_win = n_win;
        (pass == 0 ? units : fin)[i] = s;
    }
}

}  // namespace

int noalloc_setup()
{
    NoAllocScratch* p = nullptr;
    return noalloc_scratch(&p);
}

}  // namespace sz

extern "C" int szh_extract_dev_noalloc(szh_extractor* ext, const int16_t* pcm_dev, uint64_t n_samples, float* feats_dev,
                                       void* stream)
{
    using namespace sz;
    if (!ext || (n_samples && (!pcm_dev || !feats_dev))) { set_error("null argument"); return SZH_EINVAL; }
    const uint64_t nw = szh_window_count(n_samples);
    if (nw > (uint64_t)SZH_NOALLOC_MAX_WINDOWS) {
        set_error("%llu windows exceed the %d-window limit of szh_extract_dev_noalloc; use szh_plan_run_dev",
                  (unsigned long long)nw, SZH_NOALLOC_MAX_WINDOWS);
        return SZH_EINVAL;
    }
    if (nw == 0) return SZH_OK;
    if (((uintptr_t)pcm_dev & 15) != 0 || ((uintptr_t)feats_dev & 15) != 0) {
        set_error("device buffers must be 16-byte aligned");
        return SZH_EINVAL;
    }
    Extractor* e = reinterpret_cast<Extractor*>(ext);
    {   // the kernels read the extractor's tables: they must run on the device that holds them
        int cur = -1;
        SZ_HIP(hipGetDevice(&cur));
        if (cur != e->device) { set_error("extractor belongs to device %d but the current device is %d", e->device, cur); return SZH_EINVAL; }
    }
    NoAllocScratch* sc = nullptr;
    if (int rc = noalloc_scratch(&sc)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t n_win = (uint32_t)nw;
    const uint32_t n_units = (n_win + kNoAllocUnit - 1) / kNoAllocUnit, n_fin = (n_win + kSegWindows - 1) / kSegWindows;
    hipLaunchKernelGGL(single_clip_segments_kernel, dim3((n_units + 255) / 256), dim3(256), 0, s, sc->units, n_units, sc->fin,
                       n_fin, n_win);
    SZ_HIP(hipGetLastError());
    const uint32_t grid_a = (n_units + kBaseWaves - 1) / kBaseWaves;
    hipLaunchKernelGGL(mfcc_base_kernel, dim3(grid_a), dim3(kBaseThreads), kBaseLdsBytes, s, pcm_dev, sc->units, n_units,
                       e->dev, sc->base, nullptr);
    SZ_HIP(hipGetLastError());
    hipLaunchKernelGGL(finish_kernel, dim3(n_fin), dim3(kFinishThreads), 0, s, sc->base, sc->fin, n_fin, feats_dev, nullptr, 0);
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

namespace sz {

int augment_dev(const int16_t* in, uint64_t n, float gain, float noise_level, uint64_t shift, const float* noise_dev,
                uint64_t seed, uint64_t counter0, int16_t* out, hipStream_t stream)
{
    if (n == 0) return SZH_OK;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, n, gain, noise_level,
                       shift % n, noise_dev, seed, counter0, out);
    SZ_HIP(hipGetLastError());
    return SZH_OK;
}

}  // namespace sz
