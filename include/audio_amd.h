/*
 * audio_amd.h -- C ABI of libaudio_amd.so: MI355X (gfx950) native kernels for the
 * torchaudio DSP hot path (Spectrogram / MelSpectrogram / MFCC / Resample / lfilter /
 * fftconvolve).
 *
 * Boundary contract
 *   - plain C, no torch types: device pointers + sizes + a HIP stream handle (void* =
 *     hipStream_t; NULL = the legacy default stream).  The caller owns every buffer; the
 *     library allocates nothing, keeps no global mutable state and is re-entrant per
 *     (device, stream).  Kernels are enqueued asynchronously on `stream`.
 *   - every entry point returns AAMD_OK (0) or a negative AAMD_E* code;
 *     aamd_last_error() returns a thread-local message for the last failure.
 *   - all tensors are dense row-major fp32 unless stated; "rows" = the flattened leading
 *     dims of the reference's (..., time) tensors.
 *
 * What each entry point replaces in the reference (pytorch/audio, paths relative to
 * src/torchaudio/):
 *   aamd_spectrogram_f32      functional/functional.py:54-145 (F.spectrogram = pad +
 *                             torch.stft + normalisation + abs/pow), called by
 *                             transforms/_transforms.py:101-123 (Spectrogram.forward)
 *   aamd_melspectrogram_f32   transforms/_transforms.py:612-622 (MelSpectrogram.forward =
 *                             Spectrogram.forward + MelScale.forward :403-415)
 *   aamd_melspectrogram_db_f32  ... + F.amplitude_to_DB and its top_db group maximum fused
 *                             (first half of MFCC.forward, _transforms.py:692-706)
 *   aamd_melspectrogram_lognorm_f32  pipelines/rnnt_pipeline.py:16-47, 319-326 (RNN-T feature extractor: the
 *                             MelSpectrogram + transpose + gain/log + global-stats normalisation chain)
 *   aamd_kaldi_features_f32   compliance/kaldi.py:229-315 (spectrogram), :514-645 (fbank; mfcc = fbank + aamd_mfcc_dct_f32)
 *   aamd_phase_vocoder_f32    functional/functional.py:732-803 (F.phase_vocoder; T.TimeStretch, F.pitch_shift)
 *   aamd_griffinlim_update_f32  functional/functional.py:336-343 (phase update of F.griffinlim)
 *   aamd_istft_f32            functional/functional.py:148-225 (F.inverse_spectrogram -> torch.istft) and the
 *                             adjoint of the STFT for autograd of the STFT family
 *   aamd_mel_scale_f32        transforms/_transforms.py:403-415 (MelScale.forward on a
 *                             caller-supplied spectrogram)
 *   aamd_amplitude_to_db_f32  functional/functional.py:356-404 (F.amplitude_to_DB)
 *   aamd_mfcc_dct_f32         transforms/_transforms.py:692-709 (MFCC.forward after the
 *                             mel step: dB/log + top_db clamp + DCT-II matmul)
 *   aamd_resample_f32         functional/functional.py:1405-1432 (_apply_sinc_resample_kernel:
 *                             pad + strided conv1d + phase interleave + crop)
 *   aamd_resample_banded_f32  the same, banded evaluation on the matrix cores
 *   aamd_lfilter_f32          functional/filtering.py:1027-1099 (_lfilter + clamp), i.e.
 *                             DifferentiableFIR.forward :943-951 and the native IIR loop
 *                             libtorchaudio/lfilter.cpp:17-48 / iir_cuda.cu:10-35
 *                             (op torchaudio::_lfilter_core_loop, lfilter.cpp:118-124)
 *   aamd_fftconvolve_f32      functional/functional.py:2222-2258 (F.fftconvolve)
 *
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef AUDIO_AMD_H
#define AUDIO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AAMD_ABI_VERSION 7

enum {
  AAMD_OK = 0,
  AAMD_EINVAL = -1,      /* bad argument (message in aamd_last_error) */
  AAMD_EUNSUPPORTED = -2,/* valid request this build cannot serve (e.g. n_fft too large) */
  AAMD_EHIP = -3         /* a HIP runtime call / kernel launch failed */
};

/* pad modes of torch.stft(center=True) (torch/functional.py:675-680) */
enum { AAMD_PAD_REFLECT = 0, AAMD_PAD_CONSTANT = 1, AAMD_PAD_REPLICATE = 2, AAMD_PAD_CIRCULAR = 3 };

/* Framing + FFT description shared by the STFT-family entry points. */
typedef struct aamd_stft_desc {
  int64_t rows;        /* number of waveforms (flattened leading dims) */
  int64_t length;      /* samples per waveform BEFORE `pad` */
  int64_t row_stride;  /* elements between consecutive waveforms (>= length) */
  int32_t n_fft;
  int32_t hop;
  int32_t pad;         /* F.spectrogram's two-sided constant zero padding (functional.py:112-114) */
  int32_t center;      /* 0/1 */
  int32_t pad_mode;    /* AAMD_PAD_* (used when center) */
  int32_t onesided;    /* 0/1 */
  int32_t n_frames;    /* must equal 1 + (L' - n_fft)/hop, L' = length + 2*pad (+ 2*(n_fft/2) if center) */
  float   scale;       /* multiplies the complex spectrum: 1, n_fft^-1/2 ("frame_length") or 1/||w||_2 ("window") */
  float   power;       /* <= 0: complex output;  1: |X|;  2: |X|^2;  else |X|^power */
} aamd_stft_desc;

/* Banded view of a mel filterbank fb[n_freq][n_mels] (built by the host from the SAME fb
 * tensor the reference multiplies with): column m is non-zero only on rows
 * [lo[m], lo[m]+width[m]); weights[m*max_width + i] = fb[lo[m]+i][m], zero padded. */
typedef struct aamd_mel_bands {
  int32_t n_mels;
  int32_t max_width;
  const int32_t* lo;       /* device, n_mels */
  const int32_t* width;    /* device, n_mels */
  const float*   weights;  /* device, n_mels * max_width */
  /* Optional (may be NULL = identity): lane assignment for the radix-20x20 kernel, device
   * int32[ceil(n_mels/20)*20]; entry 20 r + i = the mel evaluated at lane position i in round r
   * (every mel exactly once, -1 = unused).  Results do not depend on it; it only decides which
   * LDS banks the band reads of a wavefront hit (audio_amd/_host.py: mel_lane_order). */
  const int32_t* lane_order;
  /* Optional (may be NULL): the radix-20x20 kernel's band table laid out once per filterbank by
   * aamd_mel400_table_build (device float[aamd_mel400_table_dwords(n_mels, max_width)]).  With it every workgroup of a
   * launch copies the table into LDS with one round of loads; without it each workgroup derives it from lo / width /
   * weights / lane_order (three dependent rounds, ~3 us per launch).  Results are identical. */
  const float* table400;
  /* Optional (0 = unknown): the shape of the table400 image -- 4-tap chunks of round r in nibble r, given ONLY when the
   * filterbank has exactly 4 rounds of 20 mels and every table row holds a mel.  For the signatures the library was built
   * for (HTK / Slaney 80 mels at 16 kHz: 0x4221 / 0x4211) the radix-20x20 kernel runs an instantiation whose band
   * reduction is straight-line code (- 3 %).  A signature that does not describe table400 is a caller bug: the kernel
   * checks it against the table and traps.  audio_amd/_host.py: mel400_table_signature. */
  int32_t table_sig;
} aamd_mel_bands;

/* Size (in 4-byte words) of the prebuilt band table for a filterbank, 0 if the filterbank is outside what the
 * radix-20x20 kernel serves (more than 160 mels or a band wider than 62 bins). */
int64_t aamd_mel400_table_dwords(int32_t n_mels, int32_t max_width);
/* Lay the table out (one small launch; once per filterbank and lane order, not per call).  bands->table400 is ignored. */
int aamd_mel400_table_build(const aamd_mel_bands* bands, float* table_out, void* stream);

/* Kernel-selection switches for tests and A/B measurements (never needed in production): a process-wide bit mask,
 * initialised once from the environment variables AAMD_FORCE_GENERIC / AAMD_MEL400_WIDE / AAMD_ISTFT_ATOMIC /
 * AAMD_RESAMPLE_FP32 / AAMD_FFTCONV_NO_FDL / AAMD_FFTCONV_FDL / AAMD_RESAMPLE_B32 / AAMD_MEL400_NO_POOL.
 * aamd_set_kernel_policy returns the previous mask; a negative argument only queries.  Results do not depend on the mask, only which kernel computes them. */
enum {
  AAMD_POLICY_FORCE_GENERIC = 1,  /* skip the shape-specialised kernels (radix-20x20, wave FFT, MFMA paths) */
  AAMD_POLICY_MEL400_WIDE   = 2,  /* n_fft = 400 mel epilogue: 16-byte stores through an LDS stage */
  AAMD_POLICY_ISTFT_ATOMIC  = 4,  /* inverse STFT: one atomic per contribution instead of run-based overlap-add */
  AAMD_POLICY_RESAMPLE_FP32 = 8,  /* banded resampling on v_mfma_f32_16x16x4_f32 instead of the f16 hi/lo-split MFMAs (16 x slower pipe) */
  AAMD_POLICY_FFTCONV_NO_FDL = 16, /* overlap-save: never the frequency-domain delay-line plan */
  AAMD_POLICY_FFTCONV_FDL   = 32, /* overlap-save: the COMPLEX-block delay-line plan (2) whenever the tap count allows it (cost model ignored) */
  AAMD_POLICY_FFTCONV_COMPLEX = 64, /* overlap-save: only the complex-block kernels of rounds 1-3 (plans 1 / 2), never the
                                      real-block kernel of round 4 (plan 3).  NOTE: ANY of the three FFTCONV bits selects the
                                      complex-block kernels for ALL tap counts -- also for <= 8192 taps, where plan 3 is plain
                                      overlap-save on real blocks and no delay line is involved */
  AAMD_POLICY_RESAMPLE_B32 = 128, /* f16 resampler: 4-byte LDS operand reads also where the 8-byte layout of round 5 applies
                                      (odd reduced `orig`, bands of 257 .. 448 taps: kaiser_best 44.1k -> 16k) */
  AAMD_POLICY_MEL400_NO_POOL = 256 /* n_fft = 400 kernels, builds with -DAAMD_M400_POOLS=1 only (round 5 experiment: the last tiles of
                                      every workgroup's run shared between the XCDs; bit-identical results, - 0.8 %, compiled out
                                      of the product): static tile runs per workgroup, as the product always does */
};
int         aamd_set_kernel_policy(int flags);

int         aamd_abi_version(void);
const char* aamd_last_error(void);
/* "gfx950" when the current device is an MI355X-class part; fills name (<=63 chars). */
int         aamd_device_info(char* name, int32_t name_len, int32_t* cu_count, int64_t* hbm_bytes);
/* Measurement aid (bench.py `box_calibration`; no reference counterpart): n_blocks workgroups of 256 threads each run
 * `iters` x 64 dependent fp32 FMAs per lane and record into rec[4 b .. 4 b + 3] = {shader cycles, 100 MHz wall ticks,
 * XCD id, start tick}: the shader clock the box sustains under vector-ALU load and the skew between its XCDs. */
int         aamd_box_probe(int64_t* rec, int32_t n_blocks, int32_t iters, void* stream);

/* ---- STFT family --------------------------------------------------------------------- */

/* window: device float[n_fft] (already centre zero-padded from win_length, as aten::stft does).
 * twiddle: device float[2*n_fft], twiddle[2t],[2t+1] = cos, -sin(2*pi*t/n_fft), fp64-computed.
 * out: frame-major.  power > 0: float[rows][n_frames][n_freq];  power <= 0: interleaved complex
 *      float[rows][n_frames][n_freq][2].  n_freq = onesided ? n_fft/2+1 : n_fft.
 * The reference's logical (..., freq, time) tensor is the transposed VIEW of this buffer, with
 * exactly the strides torch.stft returns. */
int aamd_spectrogram_f32(const float* wav, const float* window, const float* twiddle,
                         float* out, const aamd_stft_desc* desc, void* stream);

/* Fused STFT -> |X|^power -> banded mel.  out: float[rows][n_frames][n_mels].
 * Uses the register/LDS radix-20x20 kernel when n_fft = 400 and hop = 100, 160 or 200, center/reflect,
 * onesided, power 2; every other shape takes the generic LDS Stockham kernel -- same results.
 * (aamd_spectrogram_f32 takes the same fast kernel for that shape when power > 0.) */
int aamd_melspectrogram_f32(const float* wav, const float* window, const float* twiddle,
                            const aamd_mel_bands* bands, float* out,
                            const aamd_stft_desc* desc, void* stream);

/* The same with F.amplitude_to_DB (functional.py:390-391) fused into the epilogue:
 *   out = multiplier*log10(max(mel, amin)) - multiplier*db_multiplier,
 * and, if group_max != NULL, the running maximum of `out` over each cut-off group of
 * rows_per_group consecutive waveforms max-reduced into group_max[row / rows_per_group]
 * (float; caller pre-fills with -inf) -- the reduction top_db needs (functional.py:393-402).
 * This is the first half of MFCC.forward (transforms/_transforms.py:692-706); aamd_mfcc_dct_f32
 * with log_mode 2 is the second half. */
int aamd_melspectrogram_db_f32(const float* wav, const float* window, const float* twiddle,
                               const aamd_mel_bands* bands, float* out, const aamd_stft_desc* desc,
                               float multiplier, float amin, float db_multiplier, float* group_max,
                               int64_t rows_per_group, void* stream);

/* MFCC.forward (transforms/_transforms.py:692-709: MelSpectrogram -> amplitude_to_DB(top_db) -> DCT matmul) in ONE kernel
 * for the headline front-end (n_fft 400, hop 100 / 160 / 200, 80 mels, n_mfcc <= 48 and a multiple of 4), plus a fix-up launch:
 *   pass 0  writes out[rows][n_frames][n_mfcc] WITHOUT the top_db cut-off, max-reduces the dB values of each cut-off group
 *           into group_max (caller pre-fills with -inf) and records the smallest dB value of every 6-frame tile in tile_min;
 *   (multi-GPU: all-reduce group_max with MAX here)
 *   pass 1  redoes, clamped at group_max[g] - top_db, exactly the tiles whose minimum lies under that cut-off and counts
 *           them in fix_count (required in both passes: pass 0 resets it); every workgroup of this launch checks a strided
 *           share of the tile minima itself (its flagged tiles go to a private run of tile_list) and leaves when none is
 *           flagged -- no separate compaction kernel between the passes.  Batches in which nothing reaches the cut-off pay ~2 us for it; batches in which
 *           most tiles do should take aamd_melspectrogram_db_f32 + aamd_mfcc_dct_f32 (the caller's choice).
 * Results equal the two-kernel path's up to the rounding of the fp32 contraction order. */
typedef struct aamd_mfcc_fused {
  const float* dct_frag;   /* device float[aamd_mfcc_frag_floats()], from aamd_mfcc_frag_build */
  int32_t n_mfcc;
  int32_t pass;            /* 0 or 1 */
  float multiplier, amin, db_multiplier, top_db;   /* F.amplitude_to_DB's (functional.py:356-404) */
  float* group_max;        /* device float[ceil(rows / rows_per_group)] */
  int64_t rows_per_group;
  float* tile_min;         /* device float[aamd_mfcc_fused_tiles(desc)] */
  int32_t* fix_count;      /* device int32: pass 0 resets it, pass 1 leaves the number of tiles it redoes here */
  int32_t* tile_list;      /* device int32[aamd_mfcc_fused_tiles(desc)]: scratch of pass 1 -- each workgroup's flagged tiles (its
                              candidates are every n-th tile, so clamped tiles spread evenly however they cluster by clip) */
} aamd_mfcc_fused;
int32_t aamd_mfcc_frag_floats(void);
int64_t aamd_mfcc_fused_tiles(const aamd_stft_desc* desc);
int aamd_mfcc_fused_supported(const aamd_stft_desc* desc, const aamd_mel_bands* bands, int32_t n_mfcc);   /* 1 / 0 */
/* dct: device float[n_mels][n_mfcc] (F.create_dct, functional.py:636-667) -> the MFMA operand layout of the fused kernel */
int aamd_mfcc_frag_build(const float* dct, int32_t n_mels, int32_t n_mfcc, float* frag, void* stream);
int aamd_mfcc_fused_f32(const float* wav, const float* window, const float* twiddle, const aamd_mel_bands* bands,
                        float* out, const aamd_stft_desc* desc, const aamd_mfcc_fused* f, void* stream);

/* MelSpectrogram with the RNN-T front-end's feature post-processing fused into the epilogue
 * (pipelines/rnnt_pipeline.py:16-47, 319-326: x * gain -> _piecewise_linear_log -> (x - mean) * invstddev):
 *   out[row][t][m] = (plog(mel[row][t][m] * gain) - mean[m]) * invstddev[m]
 *   plog = the reference's two in-place masked assignments as evaluated: y <= e: y / e; e < y <= e^e: ln(y) / e;
 *   y > e^e: ln(y)   (the second mask sees the values the first one wrote)
 * out is frame-major float[rows][out_frames][n_mels] with out_frames >= n_frames; frames n_frames .. out_frames-1
 * (the pipeline's right padding) are NOT written: the caller zero-fills them.  mean / invstddev: float[n_mels].
 * Shapes outside the n_fft = 400 fast path need out_frames == n_frames. */
int aamd_melspectrogram_lognorm_f32(const float* wav, const float* window, const float* twiddle,
                                    const aamd_mel_bands* bands, float* out, const aamd_stft_desc* desc, float gain,
                                    const float* mean, const float* invstddev, int64_t out_frames, void* stream);

/* The same two kernels reading 16-bit PCM directly (what the decoder upstream of the transform produces,
 * torchaudio/_torchcodec.py -> float = int16 / 32768): the int16 -> float pass and half of the input bytes disappear.
 * wav: int16[rows][row_stride]; fold the 1 / 32768 into desc->scale.  mean == invstddev == NULL: plain mel spectrogram
 * (out_frames ignored); otherwise the RNN-T feature epilogue of aamd_melspectrogram_lognorm_f32.
 * Served for n_fft = 400, hop 160 / 200 (centre, reflect, power 2); other shapes: AAMD_EUNSUPPORTED (convert first). */
int aamd_melspectrogram_pcm16_f32(const int16_t* wav, const float* window, const float* twiddle,
                                  const aamd_mel_bands* bands, float* out, const aamd_stft_desc* desc, float gain,
                                  const float* mean, const float* invstddev, int64_t out_frames, void* stream);

/* The same, reading INTERLEAVED 16-bit PCM: pcm = int16[clips][row_stride][channels] (time-major, channels adjacent: the
 * decoder's native order, which the reference transposes to (channel, time) before any transform,
 * torchaudio/_torchcodec.py:150-152).  desc->rows = clips * channels (row r = clip r / channels, channel r % channels:
 * the output is laid out (clip, channel, frames, n_mels)); desc->length and desc->row_stride count SAMPLE TIMES per
 * channel.  channels = 1 is aamd_melspectrogram_pcm16_f32; channels = 2 is served by the same kernel (both rows of a clip
 * stage the same 32-bit (L, R) words and the gather takes the row's half-word: de-interleave, conversion and 1 / 32768 in
 * the load); any other count: AAMD_EUNSUPPORTED (transpose first). */
int aamd_melspectrogram_pcm16_interleaved_f32(const int16_t* pcm, int32_t channels, const float* window, const float* twiddle,
                                              const aamd_mel_bands* bands, float* out, const aamd_stft_desc* desc,
                                              float gain, const float* mean, const float* invstddev, int64_t out_frames,
                                              void* stream);

/* Half / bfloat16 WAVEFORMS (ABI 6; the reference accepts any floating dtype -- functional/functional.py:1413-1414, the
 * `forward` of every transform -- and returns it): wav is read as binary16 / bfloat16 and converted in the load of the
 * radix-20x20 kernel (n_fft = 400, hop 160 / 200, power 2: AAMD_EUNSUPPORTED otherwise -- the host then casts and calls the
 * float entry).  Arithmetic and `out` are float32; the caller casts the result to the input dtype.  row_stride and length in
 * SAMPLES. */
enum { AAMD_DTYPE_F16 = 1, AAMD_DTYPE_BF16 = 2 };
int aamd_melspectrogram_lowp_f32(const void* wav, int32_t wav_dtype, const float* window, const float* twiddle,
                                 const aamd_mel_bands* bands, float* out, const aamd_stft_desc* desc, void* stream);

/* Backward of the |X|^p stage of F.spectrogram (functional.py:141-145), element-wise over n bins:
 *   out = dpower * p * |X|^(p-2) * X   (interleaved complex; 0 where X = 0 and p < 2)
 * -- the spectrum-domain cotangent aamd_istft_f32(adjoint = 1) turns into d loss / d waveform.  The filterbank's backward
 * (dpower = dmel * fb^T) is aamd_mel_scale_f32 with the band table of fb^T. */
int aamd_spectrogram_grad_f32(const float* spec, const float* dpower, float* out, int64_t n, float power, void* stream);

/* Both of the above in one pass for MelSpectrogram's backward: spec_inout holds the complex STFT X
 * (float[n_vec][n_freq][2]) and receives G = (sum_m fb[k][m] dmel[v][m]) * p |X|^(p-2) X.  bands_t: band table of
 * fb^T (one band of mels per bin).  dmel: float[n_vec][n_mels]. */
int aamd_melspectrogram_grad_f32(float* spec_inout, const float* dmel, const aamd_mel_bands* bands_t, int64_t n_vec,
                                 int32_t n_freq, int32_t n_mels, float power, void* stream);

/* Kaldi-compatible front-end (compliance/kaldi.py: spectrogram :229-315, fbank :514-645; the framing and per-frame
 * conditioning of _get_window :154-217): frames of `win` samples every `shift` samples of ONE waveform, DC removal, raw
 * or windowed log-energy, pre-emphasis, window, zero padding to n_fft, power spectrum, then
 *   bands == NULL: rows float[n_frames][n_fft/2 + 1] = log(max(|X|^2, eps)), column 0 = log energy   (kaldi.spectrogram)
 *   bands != NULL: rows float[n_frames][n_cols]: column first_col + m = mel-bank energy m (log(max(., eps)) if use_log),
 *                  column energy_col (if >= 0) = log energy                                        (kaldi.fbank)
 * window: float[n_fft], the window function in the first `win` entries, zero after.  n_fft = 256 / 512 / 1024 / 2048 run on
 * the register FFT; any other EVEN n_fft (round_to_power_of_two = False: 400 at 16 kHz, 200 at 8 kHz, 1102 at 44.1 kHz ...)
 * on the mixed-radix LDS kernel of csrc/kaldi_generic.h.  Dither: the caller draws the Gaussian noise (as the reference
 * does with torch.randn) and passes it in desc->noise. */
typedef struct aamd_kaldi_desc {
  int64_t n_samples, n_frames;
  int32_t n_fft, shift, win;
  int32_t snip_edges;
  float preemphasis;
  int32_t remove_dc_offset, raw_energy;
  float energy_floor;                  /* 0: no floor */
  int32_t use_power, use_log;
  int32_t energy_col, first_col, n_cols;
  float dither;                        /* 0: off; else frames += dither * noise (kaldi.py:180-183) */
  const float* noise;                  /* device float[n_utt][n_frames][win] unit Gaussian draws (the caller's RNG), NULL if dither == 0 */
  int64_t n_utt;                       /* 0 or 1: one waveform (the reference's API); > 1: a batch of equal-length utterances */
  int64_t utt_stride;                  /* samples between utterances (>= n_samples); out is float[n_utt][n_frames][row] */
} aamd_kaldi_desc;
int aamd_kaldi_features_f32(const float* wav, const float* window, const float* twiddle, const aamd_mel_bands* bands,
                            float* out, const aamd_kaldi_desc* desc, void* stream);

/* Frames -> waveform (overlap-add).  spec: interleaved complex float[rows][n_frames][n_fft/2+1][2] (frame-major,
 * onesided); window / twiddle as for aamd_spectrogram_f32; out: float[rows][length], MUST be zero-filled by the
 * caller (contributions are accumulated with atomic adds).  desc->length = output samples per row; frame t
 * covers padded-axis samples t*hop .. t*hop+n_fft-1, which map to output samples exactly as the forward's
 * padding does (centre offset n_fft/2, pad_mode, desc->pad); samples that map nowhere are dropped.
 *   adjoint == 0: torch.istft's numerator (functional/functional.py:205-216): out += scale/n_fft * w[n] * irfft-sum;
 *                 pass pad_mode = AAMD_PAD_CONSTANT (centre trimming) and inv_envelope[length] = 1 / sum_t w^2
 *                 to get the least-squares inverse, or NULL for the bare overlap-add.
 *   adjoint != 0: the adjoint of the onesided STFT (autograd of Spectrogram & co.):
 *                 out[i] += scale * w[n] * Re sum_{k=0}^{n_fft/2} spec[t][k] e^{+2 pi i n k / n_fft}
 *                 with the forward's pad_mode, so reflected / replicated edges fold back onto the samples they copied. */
int aamd_istft_f32(const float* spec, const float* window, const float* twiddle, const float* inv_envelope,
                   float* out, const aamd_stft_desc* desc, int32_t adjoint, void* stream);

/* Phase vocoder (functional/functional.py:732-803; T.TimeStretch, F.pitch_shift): stretch a complex
 * spectrogram in time by `rate`.  Strides are in COMPLEX elements, so the reference's (rows, freq, frames)
 * tensors and this library's frame-major (rows, frames, freq) buffers are both addressable.
 * n_frames_out = ceil(n_frames_in / rate); phase_advance: float[n_freq]. */
typedef struct aamd_vocoder_desc {
  int64_t rows;
  int32_t n_freq, n_frames_in, n_frames_out;
  int64_t in_stride_row, in_stride_freq, in_stride_frame;
  int64_t out_stride_row, out_stride_freq, out_stride_frame;
  double rate;
} aamd_vocoder_desc;
int aamd_phase_vocoder_f32(const float* spec, const float* phase_advance, float* out, const aamd_vocoder_desc* desc,
                           void* stream);

/* One phase update of Griffin-Lim (functional/functional.py:336-343), element-wise over n complex values:
 *   a = rebuilt - momentum * tprev;  a /= |a| + 1e-16;  tprev = rebuilt;  next = magnitude * a
 * (`momentum` is already momentum / (1 + momentum), :302). */
int aamd_griffinlim_update_f32(const float* rebuilt, float* tprev, const float* magnitude, float* next, int64_t n,
                               float momentum, void* stream);

/* MelScale.forward on an existing spectrogram given frame-major: spec float[rows][n_frames][n_freq]
 * -> out float[rows][n_frames][n_mels]. */
int aamd_mel_scale_f32(const float* spec, const aamd_mel_bands* bands, float* out,
                       int64_t rows, int32_t n_frames, int32_t n_freq, void* stream);

/* ---- dB / MFCC ------------------------------------------------------------------------- */

/* x_db = multiplier*log10(max(x, amin)) - multiplier*db_multiplier (functional.py:390-391).
 * If group_max != NULL, also atomically max-reduces x_db of group g = i / group_size into
 * group_max[g] (float bit pattern; caller pre-fills with -inf).  out may alias x. */
int aamd_amplitude_to_db_f32(const float* x, float* out, int64_t n, float multiplier, float amin,
                             float db_multiplier, float* group_max, int64_t group_size, void* stream);

/* dB conversion and top_db clamp in one sweep, given the group maxima of a previous aamd_amplitude_to_db_f32(x, NULL, ...)
 * call (out == NULL there = maximum only): out[i] = max(dB(x[i]), group_max[i / group_size] - top_db). */
int aamd_amplitude_to_db_clamped_f32(const float* x, float* out, int64_t n, float multiplier, float amin,
                                     float db_multiplier, const float* group_max, int64_t group_size, float top_db,
                                     void* stream);

/* out[i] = max(x[i], group_max[i / group_size] - top_db)  (functional.py:393-402). */
int aamd_db_clamp_f32(const float* x, float* out, int64_t n, const float* group_max,
                      int64_t group_size, float top_db, void* stream);

/* MFCC tail on frame-major mel features mel float[n_vec][n_mels]:
 *   log_mode 0: y = 10*log10(max(mel,1e-10)) clamped at group_max[g]-top_db (top_db<0: no clamp)
 *   log_mode 1: y = log(mel + 1e-6)
 *   log_mode 2: mel already holds y (dB), only clamp
 * then out[v][k] = sum_m y[v][m] * dct[m][k]   (dct: device float[n_mels][n_mfcc]).
 * g = v / vec_per_group. */
int aamd_mfcc_dct_f32(const float* mel, const float* dct, float* out, int64_t n_vec,
                      int32_t n_mels, int32_t n_mfcc, int32_t log_mode, const float* group_max,
                      int64_t vec_per_group, float top_db, void* stream);

/* ---- Resample --------------------------------------------------------------------------- */

/* y[row][q*new + p] = sum_k kernel[p][k] * xpad[row][q*orig + k],  xpad = [0]*width ++ x ++ [0]*(width+orig),
 * for the first out_len = ceil(new*length/orig) outputs.  kernel: device float[new][taps],
 * taps = 2*width + orig.  out: float[rows][out_len]. */
int aamd_resample_f32(const float* wav, const float* kernel, float* out, int64_t rows,
                      int64_t length, int64_t row_stride, int32_t orig, int32_t new_, int32_t width,
                      int64_t out_len, void* stream);

/* Band description of the tap table per tile of 16 consecutive phases (HOST memory): every
 * non-negligible tap of phases [16t, 16t+16) lies in [tap_lo[t], tap_lo[t] + tap_span).
 * Taps outside the band are treated as zero (the caller decides what is negligible; the Python
 * host drops |h| <= 2^-40 max|h|, < 1e-9 of a full-scale output). */
typedef struct aamd_resample_bands {
  int32_t n_tiles;         /* ceil(new / 16) */
  int32_t tap_span;
  const int32_t* tap_lo;   /* host, n_tiles */
} aamd_resample_bands;

/* Same result as aamd_resample_f32, evaluated on the matrix cores over the banded tap table
 * (v_mfma_f32_16x16x4_f32, exact fp32 FMA chains).  bands == NULL, a band wider than 448 taps
 * or an `orig` whose double-buffered chunk exceeds the LDS fall back to aamd_resample_f32. */
int aamd_resample_banded_f32(const float* wav, const float* kernel, float* out, int64_t rows,
                             int64_t length, int64_t row_stride, int32_t orig, int32_t new_,
                             int32_t width, int64_t out_len, const aamd_resample_bands* bands,
                             void* stream);

/* Prepared tap fragments (ABI 7, round 5).  The binary16-split matrix-core kernels multiply with the filter's taps as packed
 * (hi, lo) binary16 fragments; formed by every workgroup in its prologue they cost 13 % of a BASELINE config-3 launch.  A filter
 * that is applied many times (T.Resample holds its `kernel` buffer, transforms/_transforms.py:966-980) prepares them once:
 *   aamd_resample_frag_bytes      size of the table for this band table (0: no matrix-core kernel serves it)
 *   aamd_resample_frag_build_f32  fills `frag` (device, 16-byte aligned, that many bytes) from the tap table, on `stream`
 *   aamd_resample_prepared_f32    = aamd_resample_banded_f32 reading `frag` (NULL: forms the fragments itself; bit-identical results)
 * `frag` must come from a build with the same kernel values, orig, new, width and band table, stream-ordered before the call
 * (or synchronised); it is read-only while calls run and independent of the kernel policy (the fp32-MFMA and scalar kernels
 * ignore it). */
int64_t aamd_resample_frag_bytes(int32_t orig, int32_t new_, const aamd_resample_bands* bands);
int     aamd_resample_frag_build_f32(const float* kernel, int32_t orig, int32_t new_, int32_t width,
                                     const aamd_resample_bands* bands, void* frag, void* stream);
int     aamd_resample_prepared_f32(const float* wav, const float* kernel, float* out, int64_t rows,
                                   int64_t length, int64_t row_stride, int32_t orig, int32_t new_,
                                   int32_t width, int64_t out_len, const aamd_resample_bands* bands,
                                   const void* frag, void* stream);

/* Sparse evaluation for ratios whose REDUCED rates are huge (F.pitch_shift / T.PitchShift: 20158 -> 16000 Hz is
 * 10079 : 8000, a tap table of 8000 x 10095 with ~36 non-negligible taps per phase; functional/functional.py:1790-1840):
 * taps_compact: device float[new][span] = kernel[p][tap_lo[p] .. tap_lo[p] + span), tap_lo: device int32[new], compacted
 * once by the host.  Same result as aamd_resample_f32 up to the taps the caller dropped. */
int aamd_resample_sparse_f32(const float* wav, const float* taps_compact, const int32_t* tap_lo, float* out, int64_t rows,
                             int64_t length, int64_t row_stride, int32_t orig, int32_t new_, int32_t width, int32_t span,
                             int64_t out_len, void* stream);

/* ---- lfilter ---------------------------------------------------------------------------- */

/* x, y: float[batch][channels][length]; a, b: device float[n_coeff_rows][n_order] with
 * n_coeff_rows = 1 (shared) or channels (per channel), lower delays first, NOT yet normalised
 * by a0 (the kernel divides, like filtering.py:1028-1029).  clamp: 0/1 -> clamp(y,-1,1) after
 * the recursion.  n_stages > 1 applies a cascade: a, b are float[n_stages][n_coeff_rows][n_order]
 * and each stage's (clamped) output feeds the next -- equal to n_stages sequential F.lfilter calls.
 * clamp = 2 clamps after the LAST stage only: one higher-order filter given as second-order sections
 * (the host layer factors orders 3 .. 8 that way when the sections reproduce the filter, see
 * audio_amd/_host.py lfilter_sos; the biquad-class kernels run ~6 x faster than the general-order scan). */
int aamd_lfilter_f32(const float* x, const float* a, const float* b, float* y, int64_t batch,
                     int32_t channels, int64_t length, int32_t n_order, int32_t n_coeff_rows,
                     int32_t n_stages, int32_t clamp, void* stream);

/* ---- fftconvolve ------------------------------------------------------------------------ */

/* Linear convolution along the last dim: out[row][n] = sum_m x[rx][m] * y[ry][n-m], full length
 * nx+ny-1, then the slice [start, start+out_len) is stored (mode crop, functional.py:2207-2219).
 * x: float[n_x_rows][nx], y: float[n_y_rows][ny]; x_row_of / y_row_of are device int64[rows] maps
 * from output row to input row (broadcasting); NULL means identity.
 * The shorter operand is treated as the taps.  <= 192 taps: tiled time-domain evaluation.  More:
 * overlap-save on a 16384-point complex FFT held in LDS (two real blocks per complex FFT, taps
 * in partitions of <= 8192; tap spectra, the twiddle table and -- for 8193 .. 32768 taps -- the
 * frequency-domain delay lines of the workgroups live in `workspace`, which must hold
 * aamd_fftconvolve_workspace() bytes, 8-byte aligned; it may be NULL when that is 0).
 * aamd_fftconvolve_plan reports what a call of that shape runs on the current device: 0 time domain, 1 overlap-save
 * with the input spectrum recomputed per tap partition, 2 overlap-save with a frequency-domain delay line (uniform
 * 8192-tap partitions, one forward + one inverse FFT per block; chosen by a cost model over rows, blocks and CUs),
 * 3 (193 .. 32768 taps, the default; 24576 until ABI 5) REAL blocks as 8192-point complex FFTs, one row per workgroup of
 * 1024 threads (csrc/fftconv_fdr.h): plain overlap-save up to 8192 taps (hop = 16385 - taps), beyond that the delay line over
 * 8192-tap partitions with up to three delayed spectra in registers. */
int64_t aamd_fftconvolve_workspace(int64_t rows, int64_t n_x_rows, int64_t n_y_rows, int64_t nx, int64_t ny);
int aamd_fftconvolve_plan(int64_t rows, int64_t nx, int64_t ny, int64_t out_len);
int aamd_fftconvolve_f32(const float* x, const float* y, float* out, int64_t rows, int64_t n_x_rows,
                         int64_t n_y_rows, int64_t nx, int64_t ny, const int64_t* x_row_of,
                         const int64_t* y_row_of, int64_t start, int64_t out_len, void* workspace,
                         void* stream);
/* The same call in two stages (ABI 6), for the case the reference's users have in practice -- many batches convolved with
 * ONE room impulse response (`T.FFTConvolve` inside an augmentation loop, functional.py:2222-2258 recomputes rfft(y) every
 * call): AAMD_FFTCONV_PREPARE writes the twiddle table and the tap spectra of the call's plan into `workspace` (two small
 * launches, ~16 us on the BASELINE config-5 shard), AAMD_FFTCONV_RUN walks the rows using a workspace prepared before.
 * Both bits = aamd_fftconvolve_f32.  A RUN-only call must see the workspace of a PREPARE call with the same taps, tap rows,
 * kernel policy and plan (aamd_fftconvolve_plan; plan 3 and plan 1 keep the workspace read-only while they run, plan 2
 * holds its delay-line ring there: one call at a time), stream-ordered after it.  PREPARE alone accepts x = out = NULL;
 * with <= 192 taps (time-domain kernel) it does nothing. */
enum { AAMD_FFTCONV_PREPARE = 1, AAMD_FFTCONV_RUN = 2 };
int aamd_fftconvolve_staged_f32(const float* x, const float* y, float* out, int64_t rows, int64_t n_x_rows,
                                int64_t n_y_rows, int64_t nx, int64_t ny, const int64_t* x_row_of,
                                const int64_t* y_row_of, int64_t start, int64_t out_len, void* workspace,
                                int32_t stages, void* stream);

/* ---- float64 ------------------------------------------------------------------------------ */

/* The reference guarantees first- AND second-order gradients of this path and checks them in float64
 * (test/torchaudio_unittest/functional/autograd_impl.py:21-35, transforms/autograd_test_impl.py:30-45); its native IIR
 * loop dispatches on double too (libtorchaudio/lfilter.cpp:62-68).  These entries have the semantics of their _f32
 * namesakes on double buffers.  They are the generic kernels (no shape-specialised fast paths): precision, not
 * throughput.  aamd_spectrogram_f64: twiddle is double[2 * n_fft]; desc->power <= 0 (complex) or > 0.
 * aamd_lfilter_f64: n_stages must be 1.  aamd_fftconvolve_f64: direct evaluation, no workspace. */
int aamd_spectrogram_f64(const double* wav, const double* window, const double* twiddle, double* out,
                         const aamd_stft_desc* desc, void* stream);
int aamd_istft_f64(const double* spec, const double* window, const double* twiddle, const double* inv_envelope,
                   double* out, const aamd_stft_desc* desc, int32_t adjoint, void* stream);
int aamd_lfilter_f64(const double* x, const double* a, const double* b, double* y, int64_t batch, int32_t channels,
                     int64_t length, int32_t n_order, int32_t n_coeff_rows, int32_t n_stages, int32_t clamp, void* stream);
int aamd_resample_f64(const double* wav, const double* kernel, double* out, int64_t rows, int64_t length,
                      int64_t row_stride, int32_t orig, int32_t new_, int32_t width, int64_t out_len, void* stream);
int aamd_fftconvolve_f64(const double* x, const double* y, double* out, int64_t rows, int64_t n_x_rows, int64_t n_y_rows,
                         int64_t nx, int64_t ny, const int64_t* x_row_of, const int64_t* y_row_of, int64_t start,
                         int64_t out_len, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AUDIO_AMD_H */
