/* deft4g.h — C ABI of libdeft4g.so, the MI355X-native (gfx950, HIP) implementation of
 * deft4j's DEFLATE stream optimiser hot path.
 *
 * Every entry point replaces one seam of the Java reference (paths under /root/reference):
 *   B/ = deft4j-base/src/main/java/com/github/NeRdTheNed/deft4j/
 *   K/ = deft4j-container/src/main/java/com/github/NeRdTheNed/deft4j/container/
 *
 * Conventions: plain pointers and sizes only; inputs are borrowed for the duration of the
 * call; buffers returned through `uint8_t**` are allocated by the library and released with
 * d4g_free().  All functions return 0 on success and a negative value on failure (text in
 * d4g_last_error()).  Nothing here has a CPU fallback: without a usable HIP device
 * d4g_init() fails and every other call returns D4G_ERR_NODEVICE.
 *
 * SCOPE: mode NONE (parse, candidate search, mergeBlocks, write) and every recompress mode are built: the zlib-family
 * compressors (JavaCompressor / JZLibCompressor at level 9), the Zopfli compressors (CafeUndZopfli FIRST / LAST / NONE,
 * JZopfli's five option sets), CompressionUtil.compress and CMDUtil's recompress loop for modes CHEAP, ZOPFLI,
 * ZOPFLI_EXTENSIVE and ZOPFLI_VERY_EXTENSIVE.  An unknown mode returns D4G_ERR_ARG — never a substitute result.
 *
 * LIMITS (reported as errors of the call, never as a wrong result): one input of 2 GiB or more; 2^32 or more back-references
 * in one batch (split it); a Zopfli master block above 8 MiB (deft4j uses 8 MiB and 1 MB).
 *
 * THREADS: d4g_init / d4g_shutdown are exclusive.  Everything else may be called from several host threads at once
 * (CompressionUtil's pool, C/CompressionUtil.java:111-117): every thread gets its own HIP streams; a d4g_batch is used by
 * one thread at a time.
 */
#ifndef DEFT4G_H
#define DEFT4G_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D4G_OK 0
#define D4G_ERR_NODEVICE (-1)
#define D4G_ERR_ARG (-2)
#define D4G_ERR_RUNTIME (-3)

/* per-stream status, mirroring Deft.optimiseDeflateStream (B/Deft.java:21-34) */
#define D4G_STREAM_CHANGED 0     /* parse ok and bits saved > 0: use the new bytes            */
#define D4G_STREAM_UNCHANGED 1   /* parse ok, nothing saved: the caller keeps its ORIGINAL array */
#define D4G_STREAM_PARSE_ERROR (-1) /* DeflateStream.parse returned false: caller keeps the original */

typedef struct d4g_batch d4g_batch;

typedef struct d4g_stats {
    double ms_upload, ms_parse, ms_optimise, ms_merge, ms_write, ms_total; /* host wall clock per phase */
    int64_t n_streams, n_blocks, n_tokens, bytes_in, bytes_decoded, bytes_out;
    int64_t rounds, kernel_launches;
    int64_t search_bytes_algorithmic; /* C_in + U + C_out summed over streams */
    double ms_search_kernels;         /* device time of the candidate-search kernels (HIP events on the library's stream) */
    double ms_parse_kernels;          /* device time of scan + probe + emit + pointer-jumping kernels */
    int64_t scan_candidates, scan_confirmed, exact_probes, jump_rounds;
    double ms_state_kernels;          /* summed device time of k_exec_state_ops launches (HIP events around each launch) */
    int64_t state_launches;
    int64_t state_tokens_per_round, state_bytes_per_round; /* summed over k_exec_state_ops launches: tokens / decoded bytes of the blocks each launch covers */
    int64_t search_lanes;             /* block groups whose level sequences run concurrently (stream lanes) */
    double ms_checksum_kernels;       /* device time of the CRC-32 / Adler-32 kernels (d4g_batch_checksums) */
    /* encoder front end (d4g_batch_create_encode): device time of the hash sort, the lazy parse and the block emit */
    double ms_lz_sort, ms_lz_parse, ms_lz_emit;
    int64_t lz_parse_passes, lz_chunks_rerun, lz_symbols;
    /* d4g_batch_run_recompress: wall clock of the encode + optimise of every compressor output (front = encoder kernels,
     * search = candidate search on the encoder outputs) and of the re-parse + optimise of the winners */
    double ms_recompress_encode, ms_recompress_encode_front, ms_recompress_encode_search, ms_recompress_reoptimise;
    int64_t recompress_outputs;         /* compressor outputs that went through the candidate search */
    int64_t recompress_outputs_pruned;  /* HUFFMAN_ONLY outputs whose entropy bound already lost: not searched */
    /* Zopfli compressors (modes ZOPFLI*): wall clock of the match tables, the block-split searches, the squeeze kernel
     * and the block choice + emission; squeeze work = sum over blocks of iterations x block bytes */
    double ms_zopfli_table, ms_zopfli_split, ms_zopfli_squeeze, ms_zopfli_emit;
    int64_t zopfli_blocks, zopfli_position_iterations;
    /* fused executor (one workgroup per block, all rounds): optimiseBlock rounds it ran, rounds handed to the level executor */
    int64_t rounds_fused, fused_fallbacks;
    /* rounds of the persistent executor that were run again by the level executor because a cross-kernel wait gave up */
    int64_t persist_fallbacks;
    /* optimiseBlock rounds of long merged blocks run by the cluster kernel (the whole device on one block) */
    int64_t rounds_cluster;
} d4g_stats;

/* Select the HIP device (one process per GPU) and create the library's stream.
 * Fails (D4G_ERR_NODEVICE) when no device is usable. */
int d4g_init(int device_index);
void d4g_shutdown(void);
/* ---- several GPUs in one process (a JVM is one process; CompressionUtil's pool, C/CompressionUtil.java:99-117, and
 * DeflateFilesContainer.optimise's stream list, K/DeflateFilesContainer.java:18-43, are what fans out over them) ----
 * A *context* is one device with its own memory pool and programs; context k = device_index[k] (a device may back more than
 * one context).  d4g_init(d) == context 0 on device d.  A batch lives on the context it was created on and every call with it
 * works there, from any host thread; calls that create batches (d4g_batch_create, the one-shot calls) use the calling thread's
 * context: 0 unless the thread chose another with d4g_set_device. */
int d4g_init_devices(int n, const int* device_index);
int d4g_device_count(void);                 /* contexts initialised */
int d4g_set_device(int context);            /* the calling thread's context from now on */
const char* d4g_last_error(void);

/* ---- batch API: K/DeflateFilesContainer.java:18-43 `optimise(List<DeflateStream>, boolean)` ----
 * Streams are independent.  create() copies the inputs to HBM; run() does all device work
 * (parse -> optimise -> [mergeBlocks] -> write) with inputs and outputs resident in HBM. */
d4g_batch* d4g_batch_create(size_t n, const uint8_t* const* in, const size_t* in_len);
d4g_batch* d4g_batch_create_on(int context, size_t n, const uint8_t* const* in, const size_t* in_len);
int d4g_batch_run(d4g_batch* b, int merge_blocks);
/* status: D4G_STREAM_*; saved_bits = DeflateStream.optimise(mergeBlocks) (B/deflate/DeflateStream.java:496);
 * out_len = bytes DeflateStream.asBytes() (:652) would return; consumed = input bytes parse() read
 * (byte-exact, K/GZFile.java:84 depends on it); size_bits_in = DeflateStream.getSizeBits() (:171) of the input. */
int d4g_batch_stream_result(d4g_batch* b, size_t i, int32_t* status, int64_t* saved_bits, size_t* out_len,
                            size_t* consumed, int64_t* size_bits_in);
/* copy stream i's re-serialised bytes (DeflateStream.write, :128-145) to host memory */
int d4g_batch_copy_output(d4g_batch* b, size_t i, uint8_t* dst, size_t cap);
/* copy stream i's decoded bytes (DeflateStream.getUncompressedData, :159-169) */
int d4g_batch_copy_decoded(d4g_batch* b, size_t i, uint8_t* dst, size_t cap, size_t* len);
/* Trailer values of stream i computed on the device from its decoded bytes: CRC-32 and ISIZE as
 * K/GZFile.java:129-145 recomputes them (java.util.zip.CRC32), Adler-32 as K/ZLibFile.java:41-51 does.
 * Valid after d4g_batch_run (or d4g_batch_parse). */
int d4g_batch_checksums(d4g_batch* b, size_t i, uint32_t* crc32, uint32_t* adler32, int64_t* isize);
/* parse + decode only (DeflateStream.parse for every stream); makes copy_decoded / checksums available */
int d4g_batch_parse(d4g_batch* b);
int d4g_batch_stats(d4g_batch* b, d4g_stats* st);
void d4g_batch_destroy(d4g_batch* b);

/* ---- encoder front end: the LZ77 compressors behind deft4j's recompress modes ----
 * One d4g_encoder_spec = one output of one Compressor object in CompressionUtil.getCompressors
 * (deft4j-compress/.../util/compression/CompressionUtil.java:44-78):
 *   D4G_ENC_JVM    JavaCompressor   (JavaCompressor.java:36-49): java.util.zip.Deflater(BEST_COMPRESSION, nowrap) = zlib level 9
 *   D4G_ENC_JZLIB  JZLibCompressor  (JZLibCompressor.java:29-41): jzlib 1.1.x level 9 (same algorithm, early block flush)
 * with strategy DEFAULT / FILTERED / HUFFMAN_ONLY.  The batch's streams are the specs' outputs, in order; several specs
 * may name the same input (its hash chains are built once).  run_encode(optimise = 0) leaves every stream exactly as the
 * encoder emits it (SingleCompressor.compressSingle); optimise = 1 also runs Deft.optimiseDeflateStream on it
 * (CompressorTask.java:29-35) without serialising and re-parsing the encoder's output.  Results come through
 * d4g_batch_stream_result / d4g_batch_copy_output: size_bits_in = bit size of the encoder's own output, saved_bits = what
 * the optimiser took off it. */
#define D4G_ENC_JVM 0
#define D4G_ENC_JZLIB 1
#define D4G_STRATEGY_DEFAULT 0
#define D4G_STRATEGY_FILTERED 1
#define D4G_STRATEGY_HUFFMAN_ONLY 2
typedef struct d4g_encoder_spec { int32_t input, encoder, strategy; } d4g_encoder_spec;
d4g_batch* d4g_batch_create_encode(size_t n_in, const uint8_t* const* raw, const size_t* raw_len, size_t n_out,
                                   const d4g_encoder_spec* spec);
int d4g_batch_run_encode(d4g_batch* b, int optimise, int merge_blocks);
/* SingleCompressor.compressSingle for n inputs with one encoder setting: out[i] is always set (release with d4g_free) */
int d4g_deflate_streams(size_t n, const uint8_t* const* raw, const size_t* raw_len, int encoder, int strategy, uint8_t** out,
                        size_t* out_len);

/* ---- recompress modes (deft4j-cmd/.../cmd/CMDUtil.java:44-50, Optimise.java `--mode`) ----
 * mode = ordinal of RecompressMode: the compressor list CompressionUtil.getCompressors builds for it (:44-78), in list
 * order JVM{DEFAULT, FILTERED, HUFFMAN_ONLY}, [JZopfli], [CafeUndZopfli], JZlib{DEFAULT, FILTERED, HUFFMAN_ONLY}.
 * `iter` = Zopfli iterations (-I of the reference CLI, default 20), used by the modes from ZOPFLI up. */
/* ---- Zopfli encoder (the recompress modes ZOPFLI / ZOPFLI_EXTENSIVE / ZOPFLI_VERY_EXTENSIVE) ----
 * MultiCafeUndZopfliCompressor.compressWithOptions (C/MultiCafeUndZopfliCompressor.java:48-52: CafeUndZopfli, master block
 * 8 << 20, BlockSplitting FIRST / LAST / NONE, `iter` iterations) and MultiJZopfliCompressor.compressWithOptions
 * (C/MultiJZopfliCompressor.java:78-85: jzopfli, master block 1000000, blocksplittingmax 15 or 0) for n inputs with one
 * option set: out[i] is a complete raw deflate stream (release with d4g_free).  splitting: D4G_ZOPFLI_SPLIT_*;
 * max_blocks: 0 = unlimited; master_block: bytes per independently encoded part (0 = the whole input; at most 8 MiB).
 * The dependencies themselves are not in the reference tree: the algorithm is the published Zopfli, pinned through
 * oracle/zopfli_oracle.c to libzopfli 1.0.3 and the reference's own asyoulik-zopfli fixture. */
#define D4G_ZOPFLI_SPLIT_FIRST 0
#define D4G_ZOPFLI_SPLIT_LAST 1
#define D4G_ZOPFLI_SPLIT_NONE 2
int d4g_zopfli_streams(size_t n, const uint8_t* const* raw, const size_t* raw_len, int iterations, int splitting, int max_blocks,
                       size_t master_block, uint8_t** out, size_t* out_len);
/* Test hooks (tests/ only): the match table of one input as Zopfli's sublen arrays — len16[i], dist16[i] and, when sublen
 * is not NULL, sublen[i * 259 + l] for l <= len16[i] — for block end `end` (0 = the input's end); and the length-limited
 * code lengths of one frequency vector (n <= 288, maxbits <= 15). */
int d4g_debug_zopfli_table(const uint8_t* raw, size_t n, size_t end, uint16_t* len16, uint16_t* dist16, uint16_t* sublen);
int d4g_debug_zopfli_code_lengths(const uint32_t* freq, int n, int maxbits, uint32_t* lengths);

#define D4G_MODE_NONE 0
#define D4G_MODE_CHEAP 1
#define D4G_MODE_ZOPFLI 2
#define D4G_MODE_ZOPFLI_EXTENSIVE 3
#define D4G_MODE_ZOPFLI_VERY_EXTENSIVE 4
/* CompressionUtil.compress(uncompressedData, threaded) (CompressionUtil.java:106-182) with useDeft = compareDeft = true
 * as CMDUtil configures it: every compressor output goes through Deft.optimiseDeflateStream(out, merge_blocks) and the
 * strict minimum by parsed bit size wins, ties to the earlier compressor in list order (the non-threaded loop :144-168 —
 * the threaded one takes completion order, which is not deterministic).  out[i] is always set; winner[i] (optional) is
 * the index of the winning output in list order. */
int d4g_compress(size_t n, const uint8_t* const* raw, const size_t* raw_len, int mode, int iter, int merge_blocks, uint8_t** out,
                 size_t* out_len, int32_t* winner);
/* CMDUtil.optimise for n raw deflate streams (CMDUtil.java:70-105): optimise; then, mode > NONE, recompress the decoded
 * bytes (d4g_compress), re-parse and optimise the winner, and take it iff its bit size is strictly smaller than the
 * optimised original's.  status[i]: D4G_STREAM_CHANGED (out[i] set: the optimised original or the grafted recompression),
 * D4G_STREAM_UNCHANGED (out[i] = NULL: keep the input), D4G_STREAM_PARSE_ERROR.  saved_bits[i] = DeflateStream.optimise's
 * return for the original; recompress_saved[i] = originalSize - recompSize when grafted, else 0. */
/* The same on a batch made by d4g_batch_create (inputs resident in HBM): afterwards d4g_batch_stream_result /
 * d4g_batch_copy_output describe the FINAL stream (the grafted recompression where it won); d4g_batch_recompress_result
 * tells which streams were grafted and by how many bits. */
int d4g_batch_run_recompress(d4g_batch* b, int mode, int iter, int merge_blocks);
int d4g_batch_recompress_result(d4g_batch* b, size_t i, int32_t* grafted, int64_t* recompress_saved);
int d4g_recompress_streams(size_t n, const uint8_t* const* in, const size_t* in_len, int mode, int iter, int merge_blocks,
                           uint8_t** out, size_t* out_len, int64_t* saved_bits, int64_t* recompress_saved, int32_t* status);

/* ---- one-shot wrappers ----
 * Deft.optimiseDeflateStream for n streams: out[i]/out_len[i] are set only when status[i] ==
 * D4G_STREAM_CHANGED (else out[i] = NULL and the caller returns its original array). */
int d4g_optimise_streams(size_t n, const uint8_t* const* in, const size_t* in_len, int merge_blocks, uint8_t** out,
                         size_t* out_len, int64_t* saved_bits, int32_t* status);
/* The same list over every initialised context (d4g_init_devices): partitioned longest-first by compressed size, one host
 * thread and one device batch per context, no exchange between devices, outputs gathered in the caller's arrays in list order
 * (the C-level twin of deft4j_amd/shard.py; with one context it is d4g_optimise_streams). */
int d4g_optimise_streams_sharded(size_t n, const uint8_t* const* in, const size_t* in_len, int merge_blocks, uint8_t** out,
                                 size_t* out_len, int64_t* saved_bits, int32_t* status);
/* Deft.getSizeBitsFallback (B/Deft.java:48-54): parsed bit length, or len*8 when the stream does not parse */
int d4g_size_bits_fallback(const uint8_t* in, size_t len, int64_t* bits);
/* DeflateStream.parse + getUncompressedData; *consumed = bytes read.  Returns D4G_ERR_ARG-style <0 only on
 * library failure; *status receives D4G_STREAM_PARSE_ERROR for a malformed stream. */
int d4g_inflate(const uint8_t* in, size_t len, uint8_t** out, size_t* out_len, size_t* consumed, int32_t* status);
void d4g_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* DEFT4G_H */
