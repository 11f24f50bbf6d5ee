// Mel front-end: extract_melspectrogram (utils/train_utils_BEAT.py:186-190) = librosa.feature.melspectrogram(
// n_fft=1024, hop=512, power=2) -> power_to_db(ref=np.max) -> float16, then the loader's column slice
// (data_loader/lmdb_loader_BEAT_full.py:229).  librosa defaults: centred frames with zero padding, periodic Hann,
// Slaney mel basis (fmin 0, fmax sr/2, slaney norm), amin 1e-10, top_db 80.
//
// kernel 1: one workgroup per (frame, clip): the 1024-sample window is staged in LDS straight from the clip
//           (coalesced 4-byte loads, hop 512 => each sample is read by two frames, L2 absorbs it), radix-2 FFT in LDS
//           (10 stages x 512 butterflies, 2 per thread), |X|^2, then the 128x513 mel projection against the transposed
//           filter table [513][128] (coalesced over the mel index).
// kernel 2: one workgroup per clip: max reduction, dB, floor, fp16 rounding, slice.
#include "common.h"
#include <hip/hip_fp16.h>
#include <math.h>

namespace {

// Two frames per transform: frame 2j is the real part and frame 2j+1 the imaginary part of ONE 1024-point complex FFT; their spectra separate
// as X0[k] = (Z[k] + conj Z[N-k]) / 2, X1[k] = (Z[k] - conj Z[N-k]) / 2i.  Halves the butterflies per frame (the first version transformed one
// real frame with a zero imaginary part).  The cross-talk between the two frames is bounded by fp32 round-off of the louder one (power floor
// ~4e-15 of its power: 60 dB below the 80 dB clamp of power_to_db).
__global__ __launch_bounds__(256) void mel_power_kernel(const float* __restrict__ audio, int n_samples, const float* __restrict__ melfb_t,
                                                        const float* __restrict__ window, const float* __restrict__ twiddle,
                                                        const int* __restrict__ band, float* __restrict__ melpow, int n_frames) {
    __shared__ float re[1024], im[1024];
    const int f0 = blockIdx.x * 2, f1 = f0 + 1, b = blockIdx.y, tid = threadIdx.x;
    const float* clip = audio + (size_t)b * n_samples;
    for (int i = tid; i < 1024; i += 256) {
        const int s0 = f0 * 512 - 512 + i, s1 = s0 + 512;
        const float wv = window[i];
        const int r = (int)(__brev((unsigned)i) >> 22);       // 10-bit reversal
        re[r] = (s0 >= 0 && s0 < n_samples) ? clip[s0] * wv : 0.f;
        im[r] = (f1 < n_frames && s1 >= 0 && s1 < n_samples) ? clip[s1] * wv : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < 10; ++s) {
        const int half = 1 << s;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = tid + u * 256;
            const int pos = j & (half - 1), i0 = ((j >> s) << (s + 1)) + pos, i1 = i0 + half;
            const int tw = pos << (9 - s);
            const float c = twiddle[2 * tw], sn = twiddle[2 * tw + 1];     // exp(-2*pi*i*tw/1024) = c + i*sn
            const float xr = re[i1], xi = im[i1];
            const float tr = xr * c - xi * sn, ti = xr * sn + xi * c;
            const float ar = re[i0], ai = im[i0];
            re[i0] = ar + tr; im[i0] = ai + ti;
            re[i1] = ar - tr; im[i1] = ai - ti;
        }
        __syncthreads();
    }
    // power spectra of the two frames, bins 0..512: thread tid takes k = tid, tid + 256 (and 512 on thread 0)
    float pa[3], pb[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int k = u < 2 ? tid + u * 256 : 512, nk = (1024 - k) & 1023;
        pa[u] = pb[u] = 0.f;
        if (u < 2 || tid == 0) {
            const float zr = re[k], zi = im[k], yr = re[nk], yi = im[nk];
            const float r0 = 0.5f * (zr + yr), i0 = 0.5f * (zi - yi);      // X0[k]
            const float r1 = 0.5f * (zi + yi), i1 = 0.5f * (yr - zr);      // X1[k]
            pa[u] = r0 * r0 + i0 * i0;
            pb[u] = r1 * r1 + i1 * i1;
        }
    }
    __syncthreads();
    re[tid] = pa[0]; re[tid + 256] = pa[1];
    im[tid] = pb[0]; im[tid + 256] = pb[1];
    if (tid == 0) { re[512] = pa[2]; im[512] = pb[2]; }
    __syncthreads();
    // mel projection: the Slaney filters are triangles, so mel m only touches bins [band[2m], band[2m+1]) (~2*513 non-zeros in total instead of
    // 128*513); thread (m = tid & 127, frame = tid >> 7) sums its band in bin order
    const int m = tid & 127, fr = tid >> 7, f = f0 + fr;
    const float* pw = fr ? im : re;
    const int b0 = band[2 * m], b1 = band[2 * m + 1];
    float s = 0.f;
    for (int k = b0; k < b1; ++k) s += melfb_t[k * 128 + m] * pw[k];
    if (f < n_frames) melpow[((size_t)b * 128 + m) * n_frames + f] = s;
}

// one workgroup of 1024 threads per clip (the clip maximum couples all its bins; with 256 threads the 64 workgroups of a 64-clip step ran 43 us)
__global__ __launch_bounds__(1024) void mel_db_kernel(const float* __restrict__ melpow, float* __restrict__ spec, int n_frames,
                                                      int out_frames) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = melpow + (size_t)b * 128 * n_frames;
    float mx = 0.f;
    for (int i = tid; i < 128 * n_frames; i += 1024) mx = fmaxf(mx, p[i]);
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
    // power_to_db(ref=np.max): 10*log10(max(amin,S)) - 10*log10(max(amin,max S)) = 10*log10(max(amin,S)/ref); the ratio
    // form makes the peak (and an all-silent clip) exactly 0 dB, as in exact arithmetic.  max(db) = 0 => floor = -top_db.
    const float inv_ref = 1.0f / fmaxf(1e-10f, mx);
    const float floor_db = -80.f;
    for (int i = tid; i < 128 * out_frames; i += 1024) {
        const int m = i / out_frames, f = i - m * out_frames;
        float db = 10.f * log10f(fmaxf(1e-10f, p[m * n_frames + f]) * inv_ref);
        db = fmaxf(db, floor_db);
        spec[((size_t)b * 128 + m) * out_frames + f] = __half2float(__float2half_rn(db));
    }
}

double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

}  // namespace

// Host tables: Slaney mel filterbank TRANSPOSED [513][128], periodic Hann [1024], twiddles (cos, -sin) [512][2],
// per-mel non-zero bin range [128][2] (begin, end).
extern "C" int eg_mel_tables(float* h_melfb_t, float* h_window, float* h_twiddle, int32_t* h_band) {
    EG_REQUIRE(h_melfb_t && h_window && h_twiddle && h_band, EG_ERR_BAD_ARG, "eg_mel_tables: null pointer");
    const int n_mels = 128, n_bins = 513;
    const double sr = 16000.0;
    double hz[130];
    const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(sr / 2);
    for (int i = 0; i < n_mels + 2; ++i) hz[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (n_mels + 1));
    for (int m = 0; m < n_mels; ++m) {
        const double enorm = 2.0 / (hz[m + 2] - hz[m]);
        for (int k = 0; k < n_bins; ++k) {
            const double fk = (sr / 2) * k / (n_bins - 1);
            const double lower = (fk - hz[m]) / (hz[m + 1] - hz[m]);
            const double upper = (hz[m + 2] - fk) / (hz[m + 2] - hz[m + 1]);
            double w = lower < upper ? lower : upper;
            if (w < 0) w = 0;
            h_melfb_t[k * 128 + m] = (float)(w * enorm);
        }
        int lo = n_bins, hi = 0;
        for (int k = 0; k < n_bins; ++k)
            if (h_melfb_t[k * 128 + m] != 0.f) { if (k < lo) lo = k; hi = k + 1; }
        if (lo > hi) lo = hi = 0;
        h_band[2 * m] = lo; h_band[2 * m + 1] = hi;
    }
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < 1024; ++i) h_window[i] = (float)(0.5 - 0.5 * cos(2.0 * pi * i / 1024.0));
    for (int t = 0; t < 512; ++t) {
        h_twiddle[2 * t] = (float)cos(2.0 * pi * t / 1024.0);
        h_twiddle[2 * t + 1] = (float)(-sin(2.0 * pi * t / 1024.0));
    }
    return EG_OK;
}

extern "C" int64_t eg_mel_workspace_bytes(int32_t batch, int32_t n_samples) {
    const int n_frames = 1 + n_samples / 512;
    return (int64_t)batch * 128 * n_frames * (int64_t)sizeof(float);
}

extern "C" int eg_melspectrogram(const float* audio, int32_t batch, int32_t n_samples, const float* d_melfb_t,
                                 const float* d_window, const float* d_twiddle, const int32_t* d_band, float* spec, int32_t out_frames,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
    EG_REQUIRE(audio && d_melfb_t && d_window && d_twiddle && d_band && spec && workspace, EG_ERR_BAD_ARG, "eg_melspectrogram: null pointer");
    EG_REQUIRE(batch > 0 && n_samples >= 512, EG_ERR_BAD_ARG, "eg_melspectrogram: batch=%d n_samples=%d", batch, n_samples);
    const int n_frames = 1 + n_samples / 512;
    EG_REQUIRE(out_frames > 0 && out_frames <= n_frames, EG_ERR_BAD_ARG, "eg_melspectrogram: out_frames=%d of %d", out_frames, n_frames);
    EG_REQUIRE(workspace_bytes >= eg_mel_workspace_bytes(batch, n_samples), EG_ERR_WORKSPACE, "eg_melspectrogram: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* melpow = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(mel_power_kernel, dim3((n_frames + 1) / 2, batch), dim3(256), 0, st, audio, n_samples, d_melfb_t, d_window, d_twiddle,
                       d_band, melpow, n_frames);
    int rc = eg_check_launch("mel_power");
    if (rc) return rc;
    hipLaunchKernelGGL(mel_db_kernel, dim3(batch), dim3(1024), 0, st, melpow, spec, n_frames, out_frames);
    return eg_check_launch("mel_db");
}
