/*
 * rcf.h -- C ABI of librcf.so: the MI355X-native wideband channelizer, NBFM discriminator and FFT
 * peak scanner that replaces the GNU Radio hot path of MattMills/radiocapture-rf.
 *
 * The reference has no FFI of its own (it is pure Python on top of GNU Radio 3.8 blocks); this ABI is
 * the boundary a maintainer binds with ctypes (INTEGRATION.md shows the stub).  Every entry point
 * names the reference interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; device pointers are passed as void* / float*.
 *   - IQ is interleaved little-endian float32 re,im ("cf32", GNU Radio gr_complex), 8 bytes/sample.
 *   - every function returns RCF_OK (0) or a negative RCF_E* code and never throws; the text of the
 *     last failure on the calling thread is rcf_last_error().
 *   - one rcf_t is bound to one HIP device and one HIP stream; calls on one handle are serialised by
 *     an internal mutex (mirrors receiver.access_lock, rc_frontend/receiver.py:48).
 *   - there is NO CPU fallback: without a gfx950 device rcf_open() fails with RCF_EHIP.
 */
#ifndef RCF_H
#define RCF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rcf rcf_t;

#define RCF_OK        0
#define RCF_EINVAL   -1   /* bad argument */
#define RCF_ENOMEM   -2   /* host or device allocation failed */
#define RCF_EHIP     -3   /* HIP runtime error / no device */
#define RCF_ENOCHAN  -4   /* unknown channel id */
#define RCF_ECAP     -5   /* block larger than the configured capacity */
#define RCF_ESTATE   -6   /* call not valid in the current state */
#define RCF_EAGAIN   -7   /* result not ready yet */
#define RCF_ERANGE   -8   /* offset outside the source's band / non-integral decimation */

/* window types: numeric values follow gnuradio.filter.firdes.WIN_* */
#define RCF_WIN_HAMMING          0
#define RCF_WIN_BLACKMAN         2
#define RCF_WIN_KAISER           4   /* takes beta */
#define RCF_WIN_BLACKMAN_HARRIS  5

/* ------------------------------------------------------------------ library / device */
const char *rcf_version(void);
const char *rcf_last_error(void);
/* number of visible HIP devices (0 when none / no driver); never fails */
int rcf_device_count(void);
/* PCI address of HIP device `device` ("0000:75:00.0") into out[cap]: what maps a HIP ordinal to its sysfs node
 * (/sys/bus/pci/devices/<addr>/numa_node, local_cpulist) -- a box may show more cards in sysfs than HIP may use.
 * RCF_EINVAL: no such device / cap too small. */
int rcf_device_pci_bus_id(int device, char *out, size_t cap);

/* ------------------------------------------------------------------ filter design (host) */
/* gnuradio.filter.firdes.low_pass_2(gain, fs, fc, tw, att_dB, window) as called at
 * rc_frontend/channel.py:33 and p25_control_demod.py:106.  Writes the float32 taps; returns the tap
 * count, or -(needed count) when cap is too small (call with taps=NULL, cap=0 to size). */
int rcf_design_low_pass_2(double gain, double fs, double fc, double tw, double att_db, int window,
                          float *taps, int cap);
/* gnuradio.fft.window.{hamming,blackman,blackmanharris}(n) (fft_vector.py:38) */
int rcf_design_window(int window, int n, float *w);
/* firdes.low_pass / firdes.high_pass (gain, fs, cutoff, transition, window, beta): tap count from the window's
 * max_attenuation (Hamming 53, Blackman 74, Blackman-harris 92, Kaiser beta/0.1102+8.7).  Used by the analog
 * voice chain: firdes.high_pass(1, rate, 300, 30, WIN_HAMMING, 6.76) (logging_receiver.py:215) and the
 * rational resampler's Kaiser low-pass.  Same return convention as rcf_design_low_pass_2. */
#define RCF_FIR_LOW_PASS   0
#define RCF_FIR_HIGH_PASS  1
int rcf_design_firdes(int kind, double gain, double fs, double fc, double tw, int window, double beta,
                      float *taps, int cap);
/* gr-filter optfir.low_pass(gain, Fs, freq1, freq2, passband_ripple_db, stopband_atten_db): remezord() length
 * estimate + Parks-McClellan exchange (pm_remez, grid density 16, 2 extra taps) -- the audio low-pass inside
 * analog.fm_demod_cf (logging_receiver.py:214).  Same return convention as rcf_design_low_pass_2. */
int rcf_design_optfir_low_pass(double gain, double fs, double freq1, double freq2, double passband_ripple_db,
                               double stopband_atten_db, float *taps, int cap);
/* analog.fm_deemph(fs, tau) (gr-analog fm_emph.py): bilinear 1-pole/1-zero section -> iir_filter_ffd taps */
int rcf_design_fm_deemph(double fs, double tau, double btaps[2], double ataps[2]);
/* rational_resampler_fff(interpolation, decimation, taps=None, fractional_bw=None) (logging_receiver.py:216-221):
 * reduces by the gcd and designs firdes.low_pass(I, I, mid, width, WIN_KAISER, 7.0) with fractional_bw 0.4.
 * Writes the reduced ratio; same return convention for the taps. */
int rcf_design_resampler(int interpolation, int decimation, int *interp_out, int *decim_out, float *taps, int cap);
/* rc_frontend/channel.py:31-33: decim = int(fs/cr)/2 (must be integral -> else RCF_ERANGE) and the
 * tap count of low_pass_2(1.0, fs, cr/2, cr/2, 20.0, WIN_HAMMING). */
int rcf_channel_params(double samp_rate, int channel_rate, int *decim, int *ntaps);
/* The same with the rule for a non-integral int(fs/cr)/2 spelled out.  rc_frontend/channel.py:31 was written for
 * Python 2, where int / int floors: the author's deployment (configs/config_denver_massive_p25.py:20,31: 10 666 666 sps,
 * receiver_split2 = False) ran 12.5 kHz channels at decim = 853 // 2 = 426, output rate 10 666 666 / 426 = 25 039 S/s.
 * Under Python 3 the same line hands GNU Radio 426.5 and the channel cannot be built.
 *   RCF_DECIM_EXACT (default)  reject (RCF_ERANGE) unless int(fs/cr) is even       -- Python 3 behaviour made explicit
 *   RCF_DECIM_FLOOR            decim = int(fs/cr) // 2                              -- what the Python 2 deployment ran
 * out_rate (may be NULL) = fs / decim: NOT 2 cr in the floored case; consumers read it from rcf_chan_info. */
#define RCF_DECIM_EXACT 0
#define RCF_DECIM_FLOOR 1
int rcf_channel_params_ex(double samp_rate, int channel_rate, int decim_rule, int *decim, int *ntaps, double *out_rate);

/* ------------------------------------------------------------------ front-end lifecycle */
/*
 * One rcf_t == one SDR source of the reference's receiver (rc_frontend/receiver.py:170-204): it owns
 * the HBM-resident wideband buffer that replaces the per-subscriber ZMQ copies of
 * `zeromq.pub_sink('ipc:///tmp/rx_source_<id>')` (receiver.py:201-202).
 *   block_capacity : max samples per push/commit            (0 -> 1<<22)
 *   hist_capacity  : samples of history kept before a block (0 -> 1<<16; must cover T-1, the PFB
 *                    prototype length and the scan FFT length)
 *   out_capacity   : per-channel output ring, samples, rounded up to a power of two (0 -> 1<<16)
 */
int rcf_open(int device, double samp_rate, double center_freq, rcf_t **out);
int rcf_open_ex(int device, double samp_rate, double center_freq, size_t block_capacity,
                size_t hist_capacity, size_t out_capacity, rcf_t **out);
int rcf_close(rcf_t *h);
/* How the channels' rotators run (freq_xlating_fir_filter_ccc's gr::blocks::rotator, rc_frontend/channel.py:35).
 * exact = 0 (default): GNU Radio's float32 increment in closed form (float64 model) -- any block size at full speed;
 *   the IQ stream is GNU Radio's up to a slowly turning common phase (7.7e-6 .. 4.3e-4 rad after 10^6 outputs),
 *   discriminator and magnitudes are not affected.
 * exact != 0: GNU Radio's own recurrence, phase *= incr in float32 with the renormalisation every 512 calls, iterated
 *   per channel on the device before each block's FIR launches: the IQ stream then carries GNU Radio's phase output for
 *   output at any stream length.  Sequential by nature (measured ~30 ns per output per block -- 0.15 ms for a 5243-output block, whatever the
 *   channel count: channels run side by side): meant for real-time
 *   block sizes.  Plain channels only -- filterbank bin taps keep the closed form (their rotator also carries the
 *   bank's own phases).  Must be called before the first channel is opened (RCF_ESTATE otherwise); the environment
 *   variable RCF_ROTATOR=exact sets it at rcf_open. */
int rcf_set_rotator(rcf_t *h, int exact);
/* Stage-2 lag (default on; RCF_S2_LAG=0 at rcf_open or on = 0 here: off).  On a handle whose block is a power-of-two
 * filterbank plus short stage-2 channels on its bins (rcf_pfb_chan_open: BASELINE configs[1]), the stage-2 launch of block
 * n is not queued behind block n's filterbank launch -- a latency-bound tail of ~19 us behind a bandwidth-bound 100 us
 * kernel -- but rides, as the first workgroups, in block n + 1's filterbank launch; every call that could observe its
 * outputs (reads, rcf_sync, channel control, timing reads, a block that cannot carry it) queues it on its own first.
 * Results are the same bits either way. */
int rcf_set_stage2_lag(rcf_t *h, int on);
/* The decimation rule rcf_chan_open / rcf_pfb_chan_open apply on this handle (rcf_channel_params_ex); default
 * RCF_DECIM_EXACT, or RCF_DECIM_FLOOR when the environment variable RCF_DECIM_FLOOR=1 is set at rcf_open. */
int rcf_set_decim_rule(rcf_t *h, int decim_rule);
/* wait until everything queued on the handle's stream has finished */
int rcf_sync(rcf_t *h);
/* the handle's hipStream_t, for callers that time the kernels with HIP events */
void *rcf_stream(rcf_t *h);
int rcf_device(rcf_t *h);

/* ------------------------------------------------------------------ measurement */
/* Per-kernel-class timing with HIP events recorded on the handle's own stream around each launch
 * (torch.cuda.Event would only see torch's stream).  Used by bench.py for roofline.achieved. */
#define RCF_T_FIR          0   /* direct xlating-FIR bank on the wideband stream, vector-FMA kernel */
#define RCF_T_PFB          1   /* polyphase filterbank kernel */
#define RCF_T_FIR_DERIVED  2   /* stage-2 / pre-filter FIRs on narrowband rings */
#define RCF_T_DISC         3   /* discriminator */
#define RCF_T_SCAN_FFT     4   /* scan FFT + log-magnitude */
#define RCF_T_SCAN_MOVSUM  5   /* scan running sum */
#define RCF_T_HISTORY      6   /* history carry-over copy */
#define RCF_T_FIR_MFMA     7   /* the same bank on the FP32 matrix cores (>= 8 channels on one source) */
#define RCF_T_AUDIO        8   /* analog voice chain (squelch/demod/de-emphasis walk, FIRs, resampler) */
#define RCF_T_TAPS         9   /* filterbank taps: tap matrix -> channel rings, rotator + discriminator fused */
#define RCF_T_COUNT        10
/* on = 0: off; 1: every class; otherwise a mask with bit (class + 1) set for each class to time -- every timed
 * launch costs two event records on the stream (~10 us of gap), so a throughput run times only what it reports.  The
 * filterbank's launch carries its two events attached to the dispatch (one barrier packet less inside the measured
 * interval; ~4 us per timed launch); RCF_TIMING_BRACKET=1 brackets it like the other classes. */
int rcf_timing_enable(rcf_t *h, int on);
/* events around every `every`-th launch of a timed class only (default 1 = every launch).  An event record is a
 * barrier packet: ~6 us of queue gap each on MI355X, two per timed launch -- at a 0.13 ms step that is 9 % of the
 * throughput being measured; a stride keeps the per-launch durations live and the stream nearly undisturbed. */
int rcf_timing_stride(rcf_t *h, int every);
/* accumulated milliseconds and launch count of one class since the last reset (syncs the stream) */
int rcf_timing_read(rcf_t *h, int what, double *total_ms, int64_t *launches, int reset);

/* ------------------------------------------------------------------ wideband ingest */
/* Host buffer in: copies n_samples H2D behind the history and runs every consumer (channels, PFB,
 * armed scan) over the new block.  Replaces the source -> pub_sink broadcast (receiver.py:201-202)
 * and every channel's sub_source (rc_frontend/channel.py:29).  Pageable or large blocks are copied on a second stream
 * (the copy of block n + 1 overlaps the kernels of block n); a block of <= 4 MiB in pinned memory (rcf_host_alloc) is
 * fetched by a kernel on the compute stream straight out of host memory -- no second stream, no cross-stream waits:
 * what a real-time block wants (RCF_RAW_DIRECT moves the threshold).  Returns once the caller's buffer has been read. */
int rcf_push_iq(rcf_t *h, const float *iq_interleaved, size_t n_samples);
/* Zero-copy ingest: *dev_ptr is where the producer (SDR DMA, generator kernel, hipMemcpy) must put
 * the next block (device memory, room for *max_samples); rcf_commit(n) then processes the n samples
 * found there.  rcf_commit without rewriting the region re-processes the resident data as the next
 * n samples of the stream (used by bench.py: inputs already resident in HBM). */
int rcf_ingest_ptr(rcf_t *h, float **dev_ptr, size_t *max_samples);
/* copy n_samples host samples to offset `at` (samples) of the pending block without processing them:
 * for drivers that deliver a block in pieces, and for pre-loading resident data */
int rcf_ingest_write(rcf_t *h, const float *iq_interleaved, size_t n_samples, size_t at);
int rcf_commit(rcf_t *h, size_t n_samples);
/* Wire-format ingest: the host hands over the SDR's native samples (2 or 4 bytes per complex sample
 * instead of 8 over PCIe) and the conversion the reference leaves to gr-osmosdr / gr-uhd on the host
 * (rc_frontend/receiver.py:74-98,170-191) runs on the GPU: x = (float(raw) - offset) * scale per
 * component, then the block is processed like rcf_push_iq.  rtl-sdr: RCF_FMT_U8, offset 127.4,
 * scale 1/128; sc16 (USRP wire, bladeRF Q11): RCF_FMT_S16, offset 0, scale 1/32768 or 1/2048.
 * A block of up to 4 MiB in pinned memory (rcf_host_alloc) is converted straight out of host memory -- one launch, no
 * staging copy: the shape of a real-time SDR block; larger ones go through a staged copy that overlaps the previous
 * block's kernels (RCF_RAW_DIRECT=<bytes> moves the threshold, 0 = always staged).  Either way the call returns once
 * the caller's buffer has been read. */
#define RCF_FMT_CF32 0   /* float32 I,Q interleaved (rcf_group_push only: rcf_push_iq is the single-front-end form) */
#define RCF_FMT_U8   1   /* unsigned 8-bit I,Q interleaved */
#define RCF_FMT_S8   2   /* signed 8-bit I,Q interleaved */
#define RCF_FMT_S16  3   /* signed 16-bit little-endian I,Q interleaved */
int rcf_push_raw(rcf_t *h, const void *iq_raw, size_t n_samples, int fmt, float scale, float offset);
/* total samples ingested so far */
int64_t rcf_samples_in(rcf_t *h);
/* Page-locked host memory for push buffers: a pinned buffer is DMA'd in place, and rcf_push_iq / rcf_push_raw copy
 * block n+1 on a separate stream while block n's kernels run (the reference's SDR driver thread filling the next
 * buffer while GNU Radio's scheduler works on the current one).  Pageable buffers work too, staged by the runtime. */
void *rcf_host_alloc(size_t bytes);
void rcf_host_free(void *p);

/* ------------------------------------------------------------------ direct ("xlat") channels */
/*
 * rcf_chan_open == channel.channel(parent_zmq_address, port, channel_rate, samp_rate, offset)
 * (rc_frontend/channel.py:18-38): decim = int(fs/cr)/2, taps = low_pass_2(1.0, fs, cr/2, cr/2, 20,
 * HAMMING), freq_xlating_fir_filter_ccc(decim, taps, offset, fs) with GNU Radio's float32 tap-phase
 * and rotator arithmetic.  Output rate fs/decim = 2*channel_rate.  The channel starts at the next
 * ingested sample with zero history (GR semantics).
 */
int rcf_chan_open(rcf_t *h, int channel_rate, double offset_hz, int *chan_id);
/* generic form: any decimation / prototype taps (e.g. the P25 69-tap pre-filter with decim 1,
 * p25_control_demod.py:106-108).  src_chan < 0: input is the wideband stream; otherwise the output of
 * PFB bin `src_chan - RCF_SRC_PFB_BIN0` (see rcf_pfb_chan_open) or of direct channel `src_chan`. */
int rcf_chan_open_taps(rcf_t *h, int src_chan, int decim, const float *taps, int ntaps,
                       double offset_hz, int *chan_id);
/* channel.set_offset (rc_frontend/channel.py:61-63): retune, keeping rotator phase and FIR history */
int rcf_chan_set_offset(rcf_t *h, int chan_id, double offset_hz);
/* channel.destroy (rc_frontend/channel.py:64-67) */
int rcf_chan_close(rcf_t *h, int chan_id);
int rcf_chan_info(rcf_t *h, int chan_id, int *decim, int *ntaps, double *out_rate, double *offset_hz);
/* samples produced so far / not yet read */
int64_t rcf_chan_produced(rcf_t *h, int chan_id);
/* index, in the channel's SOURCE stream (wideband samples for rcf_chan_open; frames of the bank / outputs of the parent
 * for chained channels and taps), of the first sample the channel sees -- everything before it counts as zero (GNU
 * Radio's zero history, the state a freshly started channel.py flowgraph is in).  The reference's data wire carries no
 * timestamps; with this a consumer can place output n at source sample start + n * decim. */
int64_t rcf_chan_start(rcf_t *h, int chan_id);
/* Non-blocking reads (return the sample count, 0 if nothing is ready, <0 on error).  read_iq is the
 * payload of the channel's zeromq.pub_sink (rc_frontend/channel.py:36); read_fm is
 * analog.quadrature_demod_cf(gain) applied to that stream (p25_control_demod.py:120-121,
 * moto_control_demod.py:105, edacs_control_demod.py:84, logging_receiver.py:233-234).  The two
 * cursors are independent.  A reader that falls more than out_capacity behind loses the oldest
 * samples (as a PUB socket at its HWM does). */
int64_t rcf_chan_read_iq(rcf_t *h, int chan_id, float *out_interleaved, size_t max_samples);
int64_t rcf_chan_read_fm(rcf_t *h, int chan_id, float gain, float *out, size_t max_samples);
/* The same reads for MANY channels behind one stream synchronisation -- what an egress pump that serves hundreds of
 * channel.py:36 PUB sockets needs per pass (a single-channel read costs a device round trip each).  what: RCF_READ_IQ
 * (out = float2[n_chans][cap_each]) or RCF_READ_FM (out = float[n_chans][cap_each], scaled by gain); counts[i] = samples
 * copied for chan_ids[i] (0: nothing new; RCF_ENOCHAN: no such channel, RCF_EINVAL: a channel listed a second time -- it
 * has one reader position per stream; the others are still served).  `out` may be
 * pinned memory (rcf_host_alloc): the copies then overlap each other. */
#define RCF_READ_IQ 0
#define RCF_READ_FM 1
int rcf_chan_read_many(rcf_t *h, int what, const int *chan_ids, int n_chans, float gain, void *out, size_t cap_each,
                       int64_t *counts);
/* P25 C4FM front half after the discriminator (p25_control_demod.py:129-133, logging_receiver.py:240-244):
 * filter.fir_filter_fff(1, taps) over quadrature_demod_cf(gain) -- e.g. the 5-tap boxcar symbol filter
 * (1/sps,)*sps.  Enabled per channel; applies from the next block on; output read with rcf_chan_read_sym
 * at the channel's rate (the sequential symbol-timing loop, op25 fsk4_demod_ff, stays on the host). */
int rcf_chan_fm_filter(rcf_t *h, int chan_id, float gain, const float *taps, int ntaps);
int64_t rcf_chan_read_sym(rcf_t *h, int chan_id, float *out, size_t max_samples);
/* drift probe of p25_control_demod.py:123-127: moving_average_ff(window, 1) * (1/window) of the
 * discriminator output (window = 10000 there) == mean of gain*fm over the last `window` samples; this is
 * the value demod_watcher hands to frontend_connector.report_offset */
int rcf_chan_fm_level(rcf_t *h, int chan_id, float gain, int window, float *level);
/* Analog NBFM voice chain behind a channel (logging_receiver.py:211-222, file_to_wav.py:109-122):
 *   analog.pwr_squelch_cc(db, alpha, 0, True)           gates (drops) samples while the smoothed power is low
 *   analog.fm_demod_cf(...) = quadrature_demod_cf(k) -> fm_deemph(rate, tau) -> fir_filter_fff(1, audio_taps)
 *   filter.fir_filter_fff(1, firdes.high_pass(...))
 *   filter.rational_resampler_fff(8000, rate)
 * The chain starts, with zero state, at the channel's next output.  The squelch and the de-emphasis IIR are
 * sequential per channel (one lane per channel walks the new samples); the FIRs and the resampler run one
 * thread per output.  Taps are the caller's (rcf/audio.py derives them as the reference's calls do). */
typedef struct rcf_audio_params {
    double squelch_db;        /* pwr_squelch_cc threshold, dB */
    double squelch_alpha;     /* its single-pole power filter */
    float quad_gain;          /* quadrature_demod_cf gain k = rate / (2 pi deviation) */
    int reserved_;
    double deemph_b[2];       /* iir_filter_ffd(btaps, ataps, False); {1,0},{1,0} = no de-emphasis */
    double deemph_a[2];
    const float *lpf_taps;    /* audio low-pass, fir_filter_fff(1, taps) */
    int n_lpf;
    int n_hpf;
    const float *hpf_taps;    /* 300 Hz high-pass */
    const float *rs_taps;     /* rational_resampler_base_fff(interpolation, decimation, taps) */
    int n_rs;
    int interpolation;
    int decimation;
    int reserved2_;
} rcf_audio_params_t;
int rcf_chan_audio_open(rcf_t *h, int chan_id, const rcf_audio_params_t *p);
int rcf_chan_audio_close(rcf_t *h, int chan_id);
/* audio samples (resampler outputs) produced so far / samples that passed the squelch (syncs the stream) */
int rcf_chan_audio_produced(rcf_t *h, int chan_id, int64_t *n_audio, int64_t *n_ungated);
/* unread audio samples, oldest first; returns the count or a negative error */
int64_t rcf_chan_read_audio(rcf_t *h, int chan_id, float *out, size_t max_samples);
/* device pointers of the channel's rings (cf32 iq ring, f32 unit-gain discriminator ring) and their
 * power-of-two capacity: sample k lives at index k & (capacity-1) */
int rcf_chan_rings(rcf_t *h, int chan_id, void **iq_ring, void **fm_ring, size_t *capacity);
/* receiver.source_offset folds demod-reported drift into the SDR centre frequency
 * (rc_frontend/receiver.py:436-475); here the same Hz shift is added to every channel's offset. */
int rcf_source_shift(rcf_t *h, double delta_hz);

/* The discriminator of EVERY bin, computed by the filterbank's own kernel (frame-major banks: 400 / 800 / 1600 / 3200 bins
 * at the reference's channel rule, rc_frontend/channel.py:31-35).  What analog.quadrature_demod_cf does behind each
 * channel's pub_sink (p25_control_demod.py:120-121) happens before the channel ever leaves the GPU:
 *     fm_k[n] = fast_atan2f(Im, Re of bin_k[n] conj(bin_k[n - 1]) x inc_k)          (gain 1; readers apply theirs)
 * into a frame-major ring fm_ring[((n - n0) & (capacity - 1)) n_bins + k] -- the arithmetic (and the bits) of a tap opened
 * with rcf_pfb_tap_open(bin, gr_phase) + rcf_chan_set_fm_only, without the tap matrix, the IQ round trip through HBM and
 * the tap_finalize pass: 8 + 8 bytes per input sample instead of 48 at 1600 bins / decimation 800.
 *   mode 1: beside the bins ring;  mode 2: INSTEAD of it (rcf_pfb_read_bin, rcf_pfb_chan_open and zero-copy readers of the
 *   bins ring then see no new frames: RCF_ESTATE);  mode 0: off again.  Takes effect with the next block.
 *   gr_phase != 0: inc_k = the float32 rotator increment GNU Radio's freq_xlating_fir_filter_ccc on bin k would carry
 *   (what rcf_pfb_tap_open(bin, 1) models); 0: the bank's exact phases (inc_k = 1).  rcf_source_shift is followed.
 * The first frame after enabling takes its predecessor from the input history (recomputed, not stored): the handle's
 * history capacity must hold (chunk + taps-per-branch x oversampling + 2) x decim + n_bins samples (RCF_ECAP otherwise). */
int rcf_pfb_fm_enable(rcf_t *h, int mode, int gr_phase);
/* bin's discriminator samples not yet read by this call, x gain -> out; a reader more than the ring behind loses the oldest */
int64_t rcf_pfb_read_fm(rcf_t *h, int bin, float gain, float *out, size_t max_samples);
/* how many chunk hand-overs inside the bank's kernel never arrived (a bounded wait gave up: those frames were demodulated
 * against a zero predecessor).  0 on a healthy device: a check for tests and health probes, not a data path. */
int64_t rcf_pfb_fm_lost(rcf_t *h);
/* zero-copy: the device ring (floats), its capacity in frames and the relative index of its first valid frame; bin k of
 * relative frame i sits at fm_ring[(i & (capacity - 1)) n_bins + k].  Order reads on rcf_stream(h) after rcf_sync / an event. */
int rcf_pfb_fm_ring(rcf_t *h, void **fm_ring, size_t *capacity_frames, int64_t *first_frame);


/* ------------------------------------------------------------------ polyphase filterbank */
/*
 * n_bins-channel PFB with prototype `taps` and decimation `decim` (n_bins % decim == 0): bin k is
 * exactly freq_xlating_fir_filter_ccc(decim, taps, k*fs/n_bins, fs) evaluated with exact phases
 * (SURVEY.md 7.2); it is the throughput path for on-grid channels and replaces the dead
 * pfb.channelizer_ccf branch of rc_frontend/receiver.py:242-261.  Output: n_bins streams at fs/decim.
 * Supported shapes (anything else: RCF_EINVAL):
 *   n_bins in {64, 128, 256, 512, 1024}, n_bins / decim in {1, 2}, up to 16 taps per branch
 *   n_bins in {400, 800, 1600, 3200},    n_bins / decim = 1 with up to 4 taps per branch, 2 with up to 2, 4 with 1
 * The second family is what makes the bins the REFERENCE's channels: with the reference's own channel filter
 * (rcf_channel_params: decim = int(fs/cr)/2, low_pass_2(1, fs, cr/2, cr/2, 20, HAMMING)) at fs = 20 Msps,
 * cr = 12.5 kHz -- decim 800, 2909 taps -- a 1600-bin bank is every 12.5 kHz-grid channel and a 3200-bin bank every
 * 6.25 kHz-grid channel rc_frontend/channel.py:31-35 could build, at the same 25 kS/s (10 Msps: 800 bins,
 * 5 Msps: 400).
 */
int rcf_pfb_open(rcf_t *h, int n_bins, int decim, const float *taps, int ntaps);
int rcf_pfb_close(rcf_t *h);
int64_t rcf_pfb_produced(rcf_t *h);
/* bin index in [0, n_bins): bin k is centred at k*fs/n_bins for k < n_bins/2, (k-n_bins)*fs/n_bins above */
int64_t rcf_pfb_read_bin(rcf_t *h, int bin, float *out_interleaved, size_t max_samples);
/* device layout of the bin outputs (complex samples; i = n & (capacity-1), n = frame index since rcf_pfb_open).
 * Power-of-two banks: tiles of 16 frames, sample n of bin k at bins_ring[(i >> 4) * tile_pitch + 16 * k + (i & 15)]
 * with tile_pitch = 16 * n_bins + 80 (a chunk of 16 frames is one contiguous run for the kernel that writes it, a bin's
 * 16 frames are one 128-byte line for whoever reads it; the padding keeps one bin's lines off a single memory channel),
 * and *pitch = 16 (frames per tile).  400 / 800 / 1600 / 3200-bin banks: ONE ring of whole frames, sample n of bin k at
 * bins_ring[i * n_bins + k], and *pitch = 0. */
int rcf_pfb_rings(rcf_t *h, void **bins_ring, size_t *capacity, size_t *pitch);
/* stage 2 on one bin: channel.py's own rule at the bin rate -- decim2 = int(bin_rate/cr)/2,
 * low_pass_2(1.0, bin_rate, cr/2, cr/2, 20, HAMMING), xlating by delta_hz -- output as a normal
 * channel id (read_iq / read_fm). */
#define RCF_SRC_PFB_BIN0 0x40000000
int rcf_pfb_chan_open(rcf_t *h, int bin, int channel_rate, double delta_hz, int *chan_id);
/* One bin AS a channel: when the bank was opened with the reference's own channel filter (rcf_channel_params
 * decimation and prototype), bin k already is channel.channel(.., channel_rate, samp_rate, k*fs/n_bins)
 * (rc_frontend/channel.py:31-35) -- the tap copies its new frames into a normal channel id (read_iq / read_fm /
 * audio / symbol filter all work), discriminator fused into the copy.  This is what the reference's dead
 * connect_channel_pfb (rc_frontend/receiver.py:343-383) was meant to do, without its second filter stage.
 * gr_phase != 0: the channel's rotator also carries the per-output phase and magnitude increment by which GNU
 * Radio's float32 rotator differs from the bank's exact phases, so the discriminator DC matches the reference's, and
 * starts at the phase GNU Radio's rotator has at a channel's first output (1, where the bin carries
 * e^{-j 2 pi k decim n0 / n_bins} at the frame n0 it is opened at): the IQ stream is that of a
 * freq_xlating_fir_filter_ccc started at the opening sample -- except that the bin comes with the filter's history
 * in it, where a new flowgraph starts from zeros. */
int rcf_pfb_tap_open(rcf_t *h, int bin, int gr_phase, int *chan_id);
/* A tap that is only ever demodulated: tap_finalize then writes its discriminator ring alone -- 4 instead of 12 bytes
 * per output; with every bin of the 1600-bin reference grid demodulated (rc_frontend/channel.py:35 +
 * p25_control_demod.py:120-121: every channel of the reference IS demodulated, in a process of its own) the finalize
 * launch moves 0.8 GB instead of 1.34 per 2^25-sample block.  The channel's IQ stream is then not available:
 * rcf_chan_read_iq / rcf_chan_read_many(RCF_READ_IQ) / rcf_chan_rings(iq) / chaining a channel or a voice chain on it
 * fail with RCF_ESTATE, and switching it on is refused (RCF_ESTATE) while something reads that stream.  on = 0 gives the
 * stream back from the next block on (the IQ read cursor skips what was never written).  Frame-major filterbank taps
 * only (RCF_EINVAL otherwise).  The discriminator path then rotates nothing either: arg(y[n] conj(y[n-1])) of the rotated
 * stream is arg(bin[n] conj(bin[n-1]) x incr) -- equal to an ordinary tap's discriminator to float32 rounding (~1e-7 rad),
 * bit-identical however the stream is cut, continuous across a switch in mid-stream (the call converts the one ring
 * sample the next block's first discriminator output is taken against; it synchronises the stream for that). */
int rcf_chan_set_fm_only(rcf_t *h, int chan_id, int on);
/* How far bin `bin` of an exact-phase bank is from GNU Radio's own channel at that offset, before any sample is seen
 * (no device needed).  freq_xlating_fir_filter_ccc (rc_frontend/channel.py:35) builds its composite taps as
 * h[i] e^{j float32(i * fwT0)}; the float32 product is rounded to 2.4-9.8e-4 rad at |offset| -> fs/2,
 * the bank's tap phases 2 pi k i / n_bins are exact.  The difference is a constant rotation (*const_phase, radians:
 * rcf_pfb_tap_open(gr_phase) puts it into the tap's rotator) plus an error filter g[i] = h[i] (e^{j (d[i] - const)} - 1)
 * whose output -- leakage of the whole wideband input, white input power P giving P * leak_l2^2 -- is what a bin tap
 * cannot reproduce.  rcf/receiver.py routes a request to the direct kernel (rcf_chan_open, GNU Radio's float32 phases
 * tap for tap) when the predicted discriminator error gain * leak_l2 * sqrt(P_wideband / P_carrier) exceeds its budget. */
int rcf_pfb_tap_leakage(double samp_rate, int n_bins, const float *taps, int ntaps, int bin, double *leak_l2,
                        double *const_phase);
/* 1 when rcf_pfb_open would accept this shape (no device needed) */
int rcf_pfb_shape_supported(int n_bins, int decim, int ntaps);

/* ------------------------------------------------------------------ scan (fft_vector.py + fft_peak_detection.py) */
/*
 * fft_vector.py:37-60: stream_to_vector(fft_len) -> fft_vcc(fft_len, forward, blackmanharris, shift)
 * -> complex_to_mag_squared -> nlog10_ff(1, fft_len, 1) -> moving_average_ff(avg_len, 1, ..) ->
 * head(n_frames) -> skiphead(n_frames-1): one float32[fft_len] vector.
 * rcf_scan_start arms the scanner at the next ingested sample; the following pushes/commits feed it;
 * rcf_scan_result returns RCF_EAGAIN until n_frames frames have been consumed, then copies the
 * vector (the content of /tmp/fft_source_<i>) to the host.  fft_len: power of two, 256 .. 1<<20.
 */
int rcf_scan_start(rcf_t *h, int fft_len, int n_frames, int avg_len);
int rcf_scan_result(rcf_t *h, float *out_spectrum);
int rcf_scan_frames_done(rcf_t *h);
/* device pointer of the finished spectrum (float32[fft_len]); valid until the next rcf_scan_start */
int rcf_scan_result_device(rcf_t *h, void **dev_spectrum);
/*
 * fft_peak_detection.py:54-72 on a float32 spectrum of n bins: data += |min(data)|; mean (sequential
 * float64); scipy.signal.find_peaks(data, width=[min_w,max_w], prominence=prominence); keep
 * data[line] > 2*mean.  Writes the surviving `line` indices in ascending order (at most cap) and
 * returns how many survived in *count (may exceed cap).  Host arithmetic in float64, bit-exact with
 * scipy.  spectrum is a HOST pointer.
 */
int rcf_find_peaks(const float *spectrum, int64_t n, double min_w, double max_w, double prominence,
                   int64_t *idx, int64_t cap, int64_t *count, double *mean_out);
/* fft_peak_detection.py:72: frequency = int(line*hz_per_bin - bandwidth/2 + center_freq) */
int64_t rcf_peak_frequency(int64_t line, double samp_rate, int64_t fft_len, double center_freq);
/* same detection run on the device-resident result of the last scan (HIP kernels: local maxima,
 * block-skipping prominence/width walks); indices land in a device int64 buffer and are also copied to
 * idx (host) when idx != NULL.  At most min(cap, 4096) indices are kept (the device sort's size); *count
 * still reports how many survived. */
int rcf_scan_find_peaks(rcf_t *h, double prominence, int64_t *idx, int64_t cap, int64_t *count,
                        double *mean_out, void **dev_idx);

/* ------------------------------------------------------------------ multi-GPU: peak-list exchange */
/*
 * One process (one rcf_t) per GPU, one SDR front-end / spectrum slice each -- the reference already runs one
 * channelizer process per SDR (systemd/radiocapture-channelizer@.service, rc_frontend/receiver.py:67-70) and the
 * only thing its fft_based_scan.sh instances have in common is the list of detected frequencies.  These calls
 * give every rank the lists of all ranks with ONE ncclAllGather over xGMI (librccl, loaded on first use):
 *   rank 0: rcf_comm_unique_id(id) -> the host passes the 128 bytes to the other ranks (any transport)
 *   every rank: rcf_comm_init(h, rank, n_ranks, id)      (ncclCommInitRank on the handle's device; collective)
 *   after a scan: rcf_allgather_peaks(h, mine, n, all, cap, counts): all[w * cap + i], i < counts[w], is rank w's
 *   i-th value (at most cap per rank are exchanged; fixed records, 8 (cap + 1) bytes per rank -- latency-bound).
 * Without rcf_comm_init (or with n_ranks = 1 and id128 = NULL) the gather is a local copy; n_ranks = 1 WITH an id
 * builds a real one-rank communicator (same RCCL calls, one GPU).  rcf_allreduce_max is the barrier +
 * max-over-ranks the benchmark needs (syncs the handle's stream, then one ncclAllReduce on a double).
 */
int rcf_comm_unique_id(void *id128);
int rcf_comm_init(rcf_t *h, int rank, int n_ranks, const void *id128);
int rcf_comm_destroy(rcf_t *h);
int rcf_comm_size(rcf_t *h);
int rcf_allgather_peaks(rcf_t *h, const int64_t *mine, int n, int64_t *all, int cap, int *counts);
int rcf_allreduce_max(rcf_t *h, double *value);

/* ------------------------------------------------------------------ groups of front-ends */
/*
 * The reference's receiver holds ALL configured SDR sources in one top block when no -i is given
 * (rc_frontend/receiver.py:67-70,170-204; ten sources per host in configs/config_denver_dev_den817.py:25-118).  A group
 * is that: G front-ends of one device whose blocks are processed by ONE launch per stage -- one conversion (wire format
 * -> cf32, the history tails dual-written on the way), one filterbank launch over the chunks of every member (members of
 * the same bank shape; the others follow one by one), one stage-2 FIR + discriminator launch per (decimation, taps)
 * class, one tap-finalize launch, one gather for the read -- instead of G of each.  A real-time block is a handful of
 * ~5 us kernels: launched per front-end, one MI355X is launch-bound at a few hundred front-ends; grouped, the PCIe link
 * is what bounds it.  Outputs are the same bits as the members run one by one (tests/test_gpu_group.py).
 *   - members must live on one device and have the same out_capacity; a handle belongs to at most one group; while it
 *     does, it runs on the group's stream (every single-handle call keeps working) and rcf_close() on it is refused.
 *   - a call processes any SUBSET of the members: n_samples[i] = 0 (or blocks[i] = NULL) skips member i this time --
 *     independent SDRs are not synchronised, a pump takes whichever blocks are complete.
 *   - rcf_group_push returns once the callers' buffers have been read; pinned buffers (rcf_host_alloc) are read in
 *     place across PCIe, pageable ones are staged by the runtime first.
 *   - on a planning failure (a ring too small, an exhausted arena) nothing is queued and no member has advanced.
 */
typedef struct rcf_group rcf_group_t;
int rcf_group_open(rcf_t *const *handles, int n, rcf_group_t **out);
/* the members stay open and go back to their own streams */
int rcf_group_close(rcf_group_t *g);
int rcf_group_size(rcf_group_t *g);
/* blocks[i]: n_samples[i] samples of member i in `fmt` (RCF_FMT_CF32 / U8 / S8 / S16; scale and offset as rcf_push_raw) */
int rcf_group_push(rcf_group_t *g, const void *const *blocks, const size_t *n_samples, int fmt, float scale, float offset);
/* rcf_commit for every member with n_samples[i] > 0: the data is already where rcf_ingest_ptr said */
int rcf_group_commit(rcf_group_t *g, const size_t *n_samples);
/* rcf_chan_read_many over channels of several members: entry j is channel chan_ids[j] of member members[j] */
int rcf_group_read_many(rcf_group_t *g, int what, const int *members, const int *chan_ids, int n, float gain, void *out,
                        size_t cap_each, int64_t *counts);
int rcf_group_sync(rcf_group_t *g);

/* ------------------------------------------------------------------ real-time pump (native thread) */
/*
 * What moves the samples in a channelizer process -- in the reference GNU Radio's scheduler threads, started by
 * receiver.start() (rc_frontend/receiver.py:271) and channel.start() (:336) -- as ONE native thread per group with no
 * interpreter in it: whenever blocks
 * of some members are complete it pushes them as one group block, gathers the new output of their subscribed channels
 * with one launch into per-channel HOST rings (pinned memory the gather kernel writes across PCIe -- no host copy), and
 * publishes the counts once the launch has finished.  Two group blocks are in flight at most: the next one is planned
 * and queued while the previous one runs.
 *   Sources are rings of whole blocks in pinned memory (rcf_host_alloc), ring_blocks[i] blocks of block_samples each:
 *   written[i] != NULL: *written[i] counts the blocks the producer (an SDR driver thread) has completed -- block k is
 *     taken from slot k % ring_blocks[i] once *written[i] > k;
 *   written == NULL or written[i] == NULL: paced by the wall clock -- block k of member i is complete at
 *     t0 + (k + 1) * block_samples / samp_rate + phase_s[i] (a recorded or synthetic stream replayed at its rate).
 *   Latency of a block = from that instant (or from the moment the pump saw the counter) to its outputs being in host
 *   memory; late = latency above one block period; overrun = the pump got to a block more than a period after it was due.
 */
typedef struct rcf_pump rcf_pump_t;
typedef struct rcf_pump_config {
    size_t block_samples;
    int fmt;                             /* RCF_FMT_* of the source rings */
    float scale, offset;
    double samp_rate;                    /* pacing rate (samples per second of every member) */
    const void *const *rings;            /* [group size] */
    const size_t *ring_blocks;           /* [group size] */
    const double *phase_s;               /* [group size] or NULL */
    const volatile uint64_t *const *written;   /* [group size] or NULL */
    int what;                            /* RCF_READ_IQ / RCF_READ_FM: what the subscribed channels deliver */
    float gain;                          /* RCF_READ_FM: quadrature_demod_cf gain */
    const int *read_members;             /* [n_read] subscribed channels: member index ... */
    const int *read_chans;               /* ... and channel id */
    int n_read;
    size_t out_ring_samples;             /* host ring per subscribed channel, rounded up to a power of two (0: 4096) */
    int64_t n_blocks;                    /* blocks per member, then the pump stops by itself (0: until rcf_pump_stop) */
    int64_t warm_blocks;                 /* first blocks of every member that the statistics do not judge */
    int max_batch;                       /* most members per group block (0: all that are ready) */
    int cpu;                             /* pin the thread to this CPU (-1: leave it to the scheduler) */
    double start_delay_s;                /* t0 = now + this */
    double batch_window_s;               /* a complete block waits up to this long for others to share its launches (0: none) */
    int rt_priority;                     /* > 0: the thread asks for SCHED_FIFO at this priority (needs CAP_SYS_NICE; refused: it runs as it is) */
    int spin_us;                         /* idle waits up to this long are spun instead of slept (a late wake-up is a late block) */
    int max_read;                        /* subscription slots (>= n_read; 0: n_read): room for rcf_pump_subscribe while it runs */
} rcf_pump_config_t;
typedef struct rcf_pump_stats {
    int64_t blocks_done;                 /* member blocks whose outputs are in host memory */
    int64_t blocks_judged, late, overruns;
    int64_t group_blocks;                /* launches of the group (batches) */
    int64_t max_batch;
    int64_t samples_out;                 /* output samples written to the host rings */
    double latency_ms_p50, latency_ms_p99, latency_ms_max;
    double host_plan_ms, host_wait_ms;   /* thread time spent planning + queueing / waiting for the device */
    double elapsed_s;
    double max_plan_ms, max_wait_ms;     /* the longest single planning + queueing of a group block / wait for the device */
    double max_sleep_overshoot_ms;       /* how much later than asked the thread ever came back from a sleep (host scheduling) */
    int64_t slow_plans, slow_waits, slow_sleeps;   /* judged region: plans > 5 ms, device waits > 5 ms, sleeps that overshot by > 2 ms */
    int rt_priority_granted;             /* 1 when the SCHED_FIFO request of rt_priority went through */
    int running;                         /* 0 once the thread has finished */
    int error;                           /* RCF_E* that stopped it (0: none) */
    /* WHY a wait or a wake-up was late (appended in round 6).  /proc/thread-self/schedstat's run_delay is the time the
     * thread was RUNNABLE but had no CPU: lateness that is run_delay is the host scheduler (other tenants' threads on the
     * CPUs this container may use -- it has a CFS quota, no CPUs of its own and no SCHED_FIFO), not the GPU or the link.
     * -1: /proc not readable. */
    int cpu;                             /* the CPU the thread is pinned to (-1: it floats over the process's mask) */
    double runq_ms_total;                /* run_delay over the judged region */
    double slow_wait_ms_total, runq_ms_in_slow_waits;    /* device waits > 5 ms: their summed length / the run_delay inside them */
    double slow_sleep_ms_total, runq_ms_in_slow_sleeps;  /* wake-ups > 2 ms late: their summed lateness / the run_delay inside them */
    int64_t involuntary_switches;        /* getrusage(RUSAGE_THREAD).ru_nivcsw over the judged region */
} rcf_pump_stats_t;
int rcf_pump_start(rcf_group_t *g, const rcf_pump_config_t *cfg, rcf_pump_t **out);
int rcf_pump_stats(rcf_pump_t *p, rcf_pump_stats_t *st);
/* samples of subscribed channel `entry` (index into read_members / read_chans) in its host ring so far */
int64_t rcf_pump_written(rcf_pump_t *p, int entry);
/* copies up to max_items new items (cf32 pairs or floats) from *cursor on; a reader that fell more than the ring behind
 * loses the oldest (as a PUB socket at its HWM does) */
int64_t rcf_pump_read(rcf_pump_t *p, int entry, int64_t *cursor, void *out, size_t max_items);
/* rcf_pump_read for n slots at once: slot entries[i] from cursors[i] on into out + i * cap_each * 8 bytes (cf32 pairs, or
 * floats in the first half of the row), counts[i] items (-1: no such slot) */
int rcf_pump_read_many(rcf_pump_t *p, const int *entries, int64_t *cursors, int n, void *out, size_t cap_each, int64_t *counts);
/* Channels come and go while a channelizer runs (rc_frontend/receiver.py:296-342 connect_channel builds and starts a
 * channel flowgraph under the running top block, :424-435 / :635-648 release and destroy it): a running pump takes a new
 * subscription -- channel chan_id of member `member`, delivered as `what` (RCF_READ_IQ / RCF_READ_FM x gain) from where
 * the channel's own reader stands (a channel nobody has read: its first output, as far as its device ring still holds
 * it) -- into a free slot.  Returns the slot (>= 0; use it as `entry` above) and in *cursor where in
 * the slot's item count the new stream begins; RCF_ECAP when all max_read slots are taken. */
int rcf_pump_subscribe(rcf_pump_t *p, int member, int chan_id, int what, float gain, int64_t *cursor);
/* chan_id = RCF_SRC_PFB_BIN0 + bin (here and in rcf_pump_config_t.read_chans) with what = RCF_READ_FM subscribes ONE BIN of the
 * member's fused discriminator ring (rcf_pfb_fm_enable: every bin of the bank demodulated inside its launch) instead of a
 * channel: delivered from the bank's next frame on, x gain -- quadrature_demod_cf behind the reference's channel at that
 * bin's frequency (p25_control_demod.py:120-121) without a tap, a tap matrix or a tap_finalize pass. */
/* frees the slot (its channel may already be closed) */
int rcf_pump_unsubscribe(rcf_pump_t *p, int entry);
/* stops the thread (if still running), waits for it, releases the pump */
int rcf_pump_stop(rcf_pump_t *p);

#ifdef __cplusplus
}
#endif
#endif /* RCF_H */
