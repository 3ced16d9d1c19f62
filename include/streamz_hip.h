/*
 * streamz_hip.h -- C ABI of libstreamz_hip.so, the MI355X (gfx950) implementation of the
 * StreamZ feature-extraction + SimpleNeuralNet path.
 *
 * The reference has no FFI layer; its boundary for this path is the public Rust API of crate
 * streamz_rs.  Every entry point below names the reference item it stands in for
 * (paths under streamz-rs/src/).  A Rust shim with the reference signatures is a 1:1
 * wrapper over these calls.
 *
 * Conventions
 *   - every function returns 0 on success and a negative SZH_E* code on failure;
 *     szh_last_error() returns a thread-local message for the last failure on this thread;
 *   - nothing throws or aborts across the boundary;
 *   - matrices are flat, row-major, caller-owned; outputs are caller-allocated and sized by
 *     a query (szh_window_count, szh_net_output_size, ...);
 *   - "host" pointers are ordinary memory; pointers in *_dev functions are device memory of
 *     the current HIP device and `stream` is a hipStream_t passed as void* (NULL = the
 *     default stream).  *_dev functions only enqueue work; they do not synchronise;
 *   - thread-safety mirrors the reference: extract / forward / identify may be called
 *     concurrently on one handle, calls that train one net must be serialised by the caller.
 *   - there is no CPU fallback: every compute call fails with SZH_ENODEVICE when no HIP
 *     device is usable.
 */
#ifndef STREAMZ_HIP_H
#define STREAMZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SZH_OK 0
#define SZH_EINVAL (-1)    /* bad argument (null pointer, shape mismatch, capacity too small) */
#define SZH_ENODEVICE (-2) /* no usable HIP device / HIP runtime error */
#define SZH_ENOMEM (-3)
#define SZH_EIO (-4)       /* file missing, unreadable or malformed */
#define SZH_EFORMAT (-5)   /* unsupported audio format (not 16-bit PCM WAV at 44.1 kHz) */

#define SZH_FEATURE_SIZE 60   /* lib.rs:30-34 FEATURE_SIZE */
#define SZH_MFCC_SIZE 20      /* lib.rs:28 */
#define SZH_WINDOW_SIZE 800   /* lib.rs:26 */
#define SZH_SAMPLE_RATE 44100 /* lib.rs:25 */

typedef struct szh_extractor szh_extractor;
typedef struct szh_plan szh_plan;
typedef struct szh_net szh_net;

/* ---- runtime ----------------------------------------------------------------------- */

/* Select the HIP device this thread's later calls use (ordinal as in hipSetDevice). */
int szh_init(int device_ordinal);
/* Number of HIP devices this process can use (0 when there is none); does not select or initialise one. */
int szh_device_count(void);
/* Wait for everything this process has enqueued on the current device (call before exit or before
 * handing the device to another library; handles stay valid and are destroyed by their own calls). */
int szh_shutdown(void);
const char *szh_last_error(void);
/* free() for buffers this library returned (szh_load_cached_features, szh_wav_read). */
void szh_free(void *p);

/* ---- feature extraction ------------------------------------------------------------- */

/* Number of 800-sample windows at hop 400 in n_samples samples: lib.rs:288-291, 317. */
uint64_t szh_window_count(uint64_t n_samples);

/* FeatureExtractor::new, lib.rs:239-258. */
int szh_extractor_create(szh_extractor **ext);
void szh_extractor_destroy(szh_extractor *ext);
/* The 26 x 401 mel filterbank the extractor uses (lib.rs:240-248), row-major. */
int szh_extractor_mel_filters(const szh_extractor *ext, float *out);

/* FeatureExtractor::extract, lib.rs:261-263 -> window_samples_with_plan, lib.rs:279-345.
 * pcm: n_samples mono 16-bit samples (host).  out: capacity_windows * 60 floats (host).
 * *n_windows receives szh_window_count(n_samples); fails with SZH_EINVAL if that exceeds
 * capacity_windows.  n_samples < 800 gives zero windows and success. */
int szh_extract(szh_extractor *ext, const int16_t *pcm, uint64_t n_samples, float *out,
                uint64_t capacity_windows, uint64_t *n_windows);

/* The file-parallel extraction loop of main.rs:500-508 in one call: file f is
 * pcm[f][0 .. n_samples[f]) and its windows go to out[f] (capacity szh_window_count each).
 * From 32 MB of PCM on, both directions are copied through pinned memory by helper threads that live for
 * the call (STREAMZ_COPY_THREADS as for szh_identify_speaker_list). */
int szh_extract_batch(szh_extractor *ext, const int16_t *const *pcm, const uint64_t *n_samples,
                      uint32_t n_files, float *const *out);

/* Same loop with the audio already resident in device memory.  A plan fixes the file layout:
 * file f occupies samples [offsets[f], offsets[f] + n_samples[f]) of one device PCM buffer;
 * every offset must be a multiple of 8 samples.  Features of all files are written
 * back to back, file f starting at window szh_plan_window_offsets()[f]. */
int szh_plan_create(szh_extractor *ext, const uint64_t *offsets, const uint64_t *n_samples,
                    uint32_t n_files, szh_plan **plan);
void szh_plan_destroy(szh_plan *plan);
uint64_t szh_plan_total_windows(const szh_plan *plan);
/* out receives n_files + 1 prefix sums of the per-file window counts. */
int szh_plan_window_offsets(const szh_plan *plan, uint64_t *out);
/* Enqueue the extraction kernels.  feats_dev: total_windows * 60 floats.  No allocation,
 * no host copy, no synchronisation. */
int szh_plan_run_dev(szh_plan *plan, const int16_t *pcm_dev, float *feats_dev, void *stream);

/* No counterpart in the reference: szh_plan_run_dev for ONE clip, for callers that cannot
 * allocate (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
 * It allocates nothing, on the host or on the device: the segment tables and the per-window
 * MFCCs that a plan keeps in its own allocations live in a fixed 1.4 MB device array that every
 * GPU the library runs on holds until the process ends.  pcm_dev: n_samples mono samples starting
 * at a 16-byte aligned device address; feats_dev: szh_window_count(n_samples) * 60 floats.  At
 * most SZH_NOALLOC_MAX_WINDOWS windows (about 148 s of audio), else SZH_EINVAL.  Only enqueues
 * work on `stream`.  All calls on one GPU share the array: calls that overlap on one GPU (two
 * streams or two threads) race; calls on one stream are safe back to back.  szh_init on the
 * device does the one-time set-up; without it the first call on a device does. */
#define SZH_NOALLOC_MAX_WINDOWS 16384
int szh_extract_dev_noalloc(szh_extractor *ext, const int16_t *pcm_dev, uint64_t n_samples, float *feats_dev,
                            void *stream);

/* ---- SimpleNeuralNet ---------------------------------------------------------------- */

/* SimpleNeuralNet::new(input, hidden1, hidden2, output), lib.rs:767-790.
 * init: NULL -> weights ~ U(-0.5, 0.5) from `seed`, biases 0 (the reference draws from an
 * unseeded thread RNG); otherwise the flat parameter vector [w1|b1|w2|b2|w3|b3] with
 * w1[input][hidden1], w2[hidden1][hidden2], w3[hidden2][output] row-major. */
int szh_net_create(uint32_t input, uint32_t hidden1, uint32_t hidden2, uint32_t output,
                   const float *init, uint64_t seed, szh_net **net);
void szh_net_destroy(szh_net *net);
/* Clone (the reference derives Clone, lib.rs:744). */
int szh_net_clone(const szh_net *net, szh_net **out);
/* dims[4] = input, hidden1, hidden2, output.  output == SimpleNeuralNet::output_size, lib.rs:792. */
int szh_net_dims(const szh_net *net, uint32_t dims[4]);
uint32_t szh_net_output_size(const szh_net *net);
uint64_t szh_net_param_count(const szh_net *net);
int szh_net_get_params(szh_net *net, float *flat);
int szh_net_set_params(szh_net *net, const float *flat);
/* SimpleNeuralNet::add_output_class, lib.rs:797-821.  new_col: hidden2 weights of the new
 * column, or NULL to draw them ~ U(-0.5, 0.5) from `seed`.  The new bias is 0. */
int szh_net_add_output_class(szh_net *net, const float *new_col, uint64_t seed);

/* SimpleNeuralNet::forward, lib.rs:880-891, for B rows at once.  x[B][input] ->
 * probs[B][output] (softmax probabilities). */
int szh_net_forward(szh_net *net, const float *x, uint64_t B, float *probs);
/* SimpleNeuralNet::embed, lib.rs:895-900 (tanh hidden layer) -> emb[B][hidden2]. */
int szh_net_embed(szh_net *net, const float *x, uint64_t B, float *emb);
/* SimpleNeuralNet::forward_embedding, lib.rs:1073-1079 (ReLU on both layers). */
int szh_net_forward_embedding(szh_net *net, const float *x, uint64_t B, float *emb);
/* SimpleNeuralNet::train, lib.rs:954-999 (one sample, scale = lr). */
int szh_net_train(szh_net *net, const float *x, const float *target, float lr);
/* SimpleNeuralNet::train_batch, lib.rs:1002-1060.  target[output] is shared by the batch;
 * B == 0 is a no-op. */
int szh_net_train_batch(szh_net *net, const float *x, uint64_t B, const float *target, float lr);

/* Optional injected random streams for szh_pretrain_from_features (the reference uses an
 * unseeded thread RNG, lib.rs:596, 123; tests inject the streams to compare step by step).
 *   perm : NULL or [epochs][n_windows], row order of each epoch after the shuffle (lib.rs:601)
 *   keep : NULL or [epochs][n_windows][60] in shuffled order; 0 = element zeroed by
 *          apply_dropout (lib.rs:119-129)
 *   seed : used for whichever stream is NULL */
typedef struct szh_rng {
    const uint32_t *perm;
    const uint8_t *keep;
    uint64_t seed;
} szh_rng;

/* pretrain_from_features, lib.rs:582-628: the per-file epoch loop (shuffle, chunks of
 * batch_size, input dropout without rescale, all-zero rows skipped, loss with the
 * pre-update weights, train_batch on the survivors).  windows[n_windows][60] (host).
 * *mean_loss = mean loss over all counted windows, 0 if none. */
int szh_pretrain_from_features(szh_net *net, const float *windows, uint64_t n_windows,
                               uint32_t target_class, uint32_t num_classes, uint32_t epochs, float lr,
                               float dropout, uint32_t batch_size, const szh_rng *rng, float *mean_loss);

/* augment, lib.rs:103-116: out[i] = clamp(pcm[(i + shift) % n] * gain + noise[i] * 32767) as i16
 * (truncating).  noise[n] holds the per-sample draws from (-noise_level, noise_level); NULL ->
 * drawn from `seed`.  shift < n_samples.  Host buffers. */
int szh_augment(const int16_t *pcm, uint64_t n_samples, float gain, float noise_level, uint64_t shift,
                const float *noise, uint64_t seed, int16_t *out);

/* Optional injected augmentation draws for szh_pretrain_network (the reference draws them from
 * an unseeded thread RNG, lib.rs:104-111).  Any member may be NULL (then drawn from rng->seed):
 *   gain[epochs], noise_level[epochs], shift[epochs] (taken modulo min(n_samples, 800)),
 *   noise[epochs][n_samples]. */
typedef struct szh_aug_draws {
    const float *gain;
    const float *noise_level;
    const uint64_t *shift;
    const float *noise;
} szh_aug_draws;

/* pretrain_network, lib.rs:348-397: per epoch the raw samples are augmented, re-extracted,
 * shuffled and trained on in chunks of batch_size, all on the device.  rng->perm is
 * [epochs][window_count(n_samples)], rng->keep [epochs][windows][60]. */
int szh_pretrain_network(szh_net *net, szh_extractor *ext, const int16_t *pcm, uint64_t n_samples,
                         uint32_t target_class, uint32_t num_classes, uint32_t epochs, float lr, float dropout,
                         uint32_t batch_size, const szh_aug_draws *aug, const szh_rng *rng, float *mean_loss);

/* train_from_feature_map, lib.rs:632-665: pretrain_from_features over files in list order
 * with num_classes = output_size; feats[f][n_windows[f]][60]; a NULL feats[f] is a file
 * missing from the map and is skipped.  *mean_loss = mean of the per-file losses. */
int szh_train_from_feature_map(szh_net *net, const float *const *feats, const uint64_t *n_windows,
                               const uint32_t *classes, uint32_t n_files, uint32_t epochs, float lr,
                               float dropout, uint32_t batch_size, uint64_t seed, float *mean_loss);

/* identify_speaker_list, lib.rs:1383-1411.  ids (capacity >= output_size) receives the
 * speakers with at least one window whose top probability is >= threshold, by vote count
 * descending (ties keep ascending id); per window the LAST maximum wins. */
int szh_identify_speaker_list(szh_net *net, szh_extractor *ext, const int16_t *pcm, uint64_t n_samples,
                              float threshold, uint32_t *ids, uint32_t capacity, uint32_t *n_ids);
/* A recording longer than two tiles is processed tile by tile (65536 windows each from host memory, 262144
 * from device memory; 8192 for nets the one-kernel forward + vote does not cover; the
 * environment variable STREAMZ_TILE_WINDOWS names another size): the tile's launch sequence
 * is captured once into a HIP graph and replayed, host PCM is staged through two device
 * buffers so that copies overlap kernels, and the votes stay on the device until the end.
 * The result does not depend on the tile size.  Host recordings of 32 MB and more are copied through
 * 32 MB of pinned memory by three helper threads that live for the call (STREAMZ_COPY_THREADS sets
 * their number, 0 = plain pageable copies): 50 instead of 28 GB/s of PCM.
 *
 * szh_identify_speaker_list_dev is the same function (lib.rs:1383-1411) for PCM that already
 * lies in device memory (16-byte aligned); tile_windows 0 selects the default, and counts
 * (optional, output_size entries) receives the per-speaker vote counts. */
int szh_identify_speaker_list_dev(szh_net *net, szh_extractor *ext, const int16_t *pcm_dev, uint64_t n_samples,
                                  float threshold, uint32_t tile_windows, uint32_t *ids, uint32_t capacity,
                                  uint32_t *n_ids, uint64_t *counts);
/* Pooled embedding of feature rows, unit norm (normalize, lib.rs:132-139), out[hidden2]:
 *   mode 0  extract_embedding_from_features, lib.rs:1450-1471: mean of forward_embedding
 *   mode 1  median_embedding_from_features, lib.rs:1474-1495: per-dimension median of forward_embedding
 *   mode 2  per-dimension median of embed (the pooling of extract_embedding, lib.rs:1433-1446)
 * No windows give the zero vector.  The medians are exact order statistics (radix selection on the
 * device); the mean is accumulated in f64. */
int szh_net_pooled_embedding(szh_net *net, const float *feats, uint64_t n_windows, int mode, float *out);
/* extract_embedding, lib.rs:1418-1447: extract, embed every window, per-dimension median, unit norm. */
int szh_extract_embedding(szh_net *net, szh_extractor *ext, const int16_t *pcm, uint64_t n_samples, float *out);

/* identify_speaker, lib.rs:1285-1303: the class whose window probabilities sum highest (the last
 * maximum wins; 0 for a net without outputs).  sums (optional, output_size entries) receives the
 * per-class sums.  Sums and window counts are exact in f64 on the device and compared as f32; the
 * reference adds f32 values in window order (and its f32 window count stops at 2^24), so a decision
 * can differ only where the reference's own rounding decides.  Training calls and the *_dev calls on
 * one net must be serialised by the caller; *_dev calls need the handle's device current. */
int szh_identify_speaker(szh_net *net, szh_extractor *ext, const int16_t *pcm, uint64_t n_samples, uint32_t *id,
                         float *sums);
/* identify_speaker_with_threshold, lib.rs:1307-1343: *id = the arg-max of the sums if
 * sum / windows >= threshold, else -1 (None); -1 as well for a net with at most one output or a
 * sample without a window. */
int szh_identify_speaker_with_threshold(szh_net *net, szh_extractor *ext, const int16_t *pcm, uint64_t n_samples,
                                        float threshold, int64_t *id);
/* identify_speaker_with_threshold_feats, lib.rs:1346-1377: the same on feature rows. */
int szh_identify_speaker_with_threshold_feats(szh_net *net, const float *feats, uint64_t n_windows, float threshold,
                                              int64_t *id, float *sums);
/* The same vote on feature rows already extracted (loop body of lib.rs:1391-1402).
 * counts (optional, output_size entries) receives the per-speaker vote counts. */
int szh_identify_from_features(szh_net *net, const float *feats, uint64_t n_windows, float threshold,
                               uint32_t *ids, uint32_t capacity, uint32_t *n_ids, uint64_t *counts);

/* ---- device-resident training / inference (multi-GPU building blocks) --------------- */
/* One optimiser step is: zero_grad, accumulate over any number of feature blocks (here or on
 * other ranks), all-reduce the gradient vector across ranks (caller, e.g. RCCL), apply.
 * The gradient vector is param_count floats laid out like the parameters, followed by two
 * floats: the summed loss and the count of live rows. */
uint64_t szh_net_grad_len(const szh_net *net); /* param_count + 2 */
float *szh_net_grad_dev(szh_net *net);
int szh_net_zero_grad_dev(szh_net *net, void *stream);
/* Forward + backward of n rows with per-row labels, weights frozen, sums added into the
 * gradient vector (same arithmetic as the body of lib.rs:1013-1045 per row).
 * dropout > 0 zeroes input elements with that probability from a counter-based generator
 * keyed by (seed, row index); rows that end all-zero are skipped (lib.rs:607-609). */
int szh_net_accumulate_dev(szh_net *net, const float *feats_dev, const int32_t *labels_dev, uint64_t n,
                           float dropout, uint64_t seed, void *stream);
/* params -= (lr / live_rows) * grad, lib.rs:1047-1059, reading live_rows from the vector's
 * tail (so after an all-reduce it is the global count).  No-op when live_rows == 0. */
int szh_net_apply_grad_dev(szh_net *net, float lr, void *stream);
int szh_net_forward_dev(szh_net *net, const float *feats_dev, uint64_t n, float *probs_dev, void *stream);
/* Arithmetic of the large batched products (hidden1 x hidden2 layer forward, its delta and its
 * weight gradient; the contractions of lib.rs:882-886 and 1034-1044 over a batch of windows).
 * SZH_GEMM_F32: f32 operands on v_mfma_f32_32x32x2_f32 (the default).  SZH_GEMM_BF16X3: every f32
 * operand split exactly into three bf16 values, six bf16 matrix instructions per product term,
 * f32 accumulation: the bf16 matrix-core path at f32 accuracy.  Used when hidden1 and hidden2
 * are multiples of 256 and a call has at least 1024 rows, otherwise the f32 kernels run.
 * A new net takes its mode from the environment variable STREAMZ_GEMM ("bf16x3" or "f32").
 * No counterpart in the reference, whose arithmetic is scalar f32. */
#define SZH_GEMM_F32 0
#define SZH_GEMM_BF16X3 1
int szh_net_set_gemm(szh_net *net, int mode);
int szh_net_get_gemm(const szh_net *net);
/* Votes of identify_speaker_list over device-resident feature rows: counts_dev[output]
 * (uint64) is incremented; the caller zeroes it and orders the result. */
int szh_net_vote_dev(szh_net *net, const float *feats_dev, uint64_t n, float threshold,
                     unsigned long long *counts_dev, void *stream);

/* ---- the collective of a sharded epoch (RCCL over xGMI) --------------------------------- */
/* No counterpart in the reference, which trains in one process (lib.rs:632-665, the loop the
 * shards split).  One process per GPU: each calls szh_init(local rank), builds the same net,
 * accumulates on its own files, and the ranks sum their gradient vectors in place before the
 * identical apply step.  librccl.so.1 is opened by the first of these calls, not at load time.
 *
 * szh_comm_unique_id: rank 0 draws the 128-byte id (ncclGetUniqueId) and hands it to the other
 * ranks by any host-side channel (pipe, file, environment).
 * szh_comm_create: collective over the `world` ranks (blocks until all have called), on the
 * current device.  world == 1 with id == NULL makes a communicator whose calls do nothing. */
/* szh_shard_ranges: how the file list of main.rs:500-508 / lib.rs:643 is cut over `world` ranks: contiguous
 * ranges in list order, balanced on window count.  n_windows[n_files] are the files' window counts; rank r
 * takes files [out[r], out[r + 1]), out has world + 1 entries, out[0] = 0, out[world] = n_files.  Boundary r
 * is the file index whose window prefix sum lies nearest to r / world of the total (the smaller index on a
 * tie; never before the previous boundary), so ranks may receive no file when there are fewer files than
 * ranks.  Pure host arithmetic: every rank computes the same cut without talking to the others. */
int szh_shard_ranges(const uint64_t *n_windows, uint32_t n_files, uint32_t world, uint32_t *out);

/* The transport of a communicator is fixed by its id.  szh_comm_unique_id draws an RCCL id unless the
 * environment variable STREAMZ_COMM is "host": then the id names a shared-memory segment and the sums of
 * that communicator are formed on the host (device -> shared memory, added in rank order, identical bits
 * on every rank, -> device).  That transport is for ranks that share a GPU, where RCCL refuses to form a
 * communicator (rehearsals of a multi-rank run on a smaller machine); vectors of up to 16 M floats. */
#define SZH_COMM_ID_BYTES 128
typedef struct szh_comm szh_comm;
int szh_comm_unique_id(uint8_t id[SZH_COMM_ID_BYTES]);
int szh_comm_create(const uint8_t id[SZH_COMM_ID_BYTES], int rank, int world, szh_comm **comm);
void szh_comm_destroy(szh_comm *comm);
int szh_comm_rank(const szh_comm *comm);
int szh_comm_world(const szh_comm *comm);
/* Sum over ranks of the gradient vector (szh_net_grad_len floats at szh_net_grad_dev, the loss
 * sum and live-row count in its tail included), in place, enqueued on `stream` behind the
 * accumulate kernels: no staging copy.  Every rank must call it with a net of the same shape. */
int szh_net_allreduce_grad_dev(szh_net *net, szh_comm *comm, void *stream);
/* The parameters of rank `root` replace those of every other rank (start of a sharded run). */
int szh_net_broadcast_params_dev(szh_net *net, szh_comm *comm, int root, void *stream);
/* Epoch-synchronous data-parallel training of one rank's shard (the schedule of the task's multi-GPU
 * configurations: files sharded over the ranks, one all-reduce of the weight gradients per epoch).
 * feats[n][60] and labels[n] are this rank's windows in host memory (n may be 0).  Per epoch, on every
 * rank: zero_grad, accumulate over the shard (dropout keyed by (seed + epoch, row); all-zero rows do not
 * count, lib.rs:607-609), all-reduce over comm (NULL or a one-rank comm: none), apply with lr.
 * *mean_loss = mean cross-entropy over all ranks' live rows in the last epoch.  Every rank must make
 * the same call with the same epochs / lr / dropout / seed and a net holding the same parameters.
 * Unlike train_from_feature_map (lib.rs:632-665), which steps through the files one after the other in
 * batches of 8, this takes one optimiser step per epoch over all windows. */
int szh_net_train_epochs_sharded(szh_net *net, szh_comm *comm, const float *feats, const int32_t *labels, uint64_t n,
                                 uint32_t epochs, float lr, float dropout, uint64_t seed, float *mean_loss);

/* ---- model metadata and persistence -------------------------------------------------- */

/* record_training_file, lib.rs:855-862; file_lists, lib.rs:865-867. */
int szh_net_record_training_file(szh_net *net, uint32_t speaker, const char *path);
/* Newline-joined paths of one speaker; returns the byte length (without NUL) or a negative
 * error; copies at most capacity-1 bytes and NUL-terminates when buf != NULL. */
int64_t szh_net_file_list(const szh_net *net, uint32_t speaker, char *buf, uint64_t capacity);
/* set_dataset_specs, lib.rs:823-826. */
int szh_net_set_dataset_specs(szh_net *net, uint32_t sample_rate, uint16_t bits);
/* set_embeddings / embeddings, lib.rs:870-877: n rows of (vector[dim], mean_sim, std_sim). */
int szh_net_set_embeddings(szh_net *net, const float *vectors, const float *mean_sims,
                           const float *std_sims, uint32_t n, uint32_t dim);
int szh_net_get_embeddings(const szh_net *net, float *vectors, float *mean_sims, float *std_sims,
                           uint32_t *n, uint32_t *dim);

/* SimpleNeuralNet::save / load, lib.rs:1081-1130 / 1132-1281: model.npz, an uncompressed zip
 * of .npy members (w1, b1, w2, b2, sample_rate, bits, num_speakers, w3_i / b3_i per class
 * (1-based), speaker_i_files (0-based), speaker_embeddings, speaker_mean_sims,
 * speaker_std_sims). */
int szh_net_save(szh_net *net, const char *path);
int szh_net_load(const char *path, szh_net **net);

/* load_cached_features, lib.rs:558-579: feature_cache/<path with / and \ replaced by _>.npy
 * ((n, 60) float32, C order); on a miss the audio is read (16-bit PCM WAV, 44.1 kHz,
 * channels averaged, lib.rs:517-529, 172-183), extracted and the cache written (write
 * errors ignored, empty results not written).  *feats is malloc'd; release with szh_free. */
int szh_load_cached_features(const char *path, szh_extractor *ext, float **feats, uint64_t *n_windows);

/* WAV ingest alone (load_and_resample_file, lib.rs:509-529, wav arm at 44.1 kHz):
 * *pcm is malloc'd mono int16; release with szh_free. */
int szh_wav_read(const char *path, int16_t **pcm, uint64_t *n_samples);

/* Header of a WAV file without reading its samples: the number of mono sample frames szh_wav_read would
 * return, the sample rate and the channel count (what hound's WavReader::duration / spec give the
 * reference, lib.rs:517-518).  Host only.  Lets a rank size its share of a file list before decoding. */
int szh_wav_info(const char *path, uint64_t *n_frames, uint32_t *sample_rate, uint16_t *channels);

/* ---- measurement ----------------------------------------------------------------------- */
/* No counterpart in the reference.  When enabled, every kernel launch of the extraction and
 * network paths is bracketed by a HIP event pair on its launch stream.  szh_profile_report
 * synchronises the device, writes one line "tag launches total_ms" per kernel tag seen since
 * the previous report into buf (NUL-terminated, truncated to capacity) and returns the full
 * text length.  Tags: mfcc_base, finish, prep, fwd1, fwd2, fwd3, softmax, delta2, delta1,
 * dw3, dw2, dw1, bias, reduce, apply, vote. */
int szh_profile_enable(int on);
int64_t szh_profile_report(char *buf, uint64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* STREAMZ_HIP_H */
