package com.github.NeRdTheNed.deft4j;

/**
 * JNI binding of libdeft4g.so (the MI355X implementation of deft4j's hot path); shim: jni/deft4g_jni.c.
 * Every method takes a whole list and makes one device batch of it.
 *
 * Drop-in points in deft4j (see INTEGRATION.md):
 *   Deft.optimiseDeflateStream / getSizeBitsFallback  -> optimiseStreams / sizeBitsFallback
 *   DeflateFilesContainer.optimise(List, boolean)     -> optimiseStreams (one call for the container's streams)
 *   CompressionUtil.getCompressors                    -> GpuCompressor (a SingleCompressor over deflateStreams)
 *   CompressionUtil.compress / CMDUtil.optimise       -> compress / recompressStreams
 */
public final class NativeDeft {
    /** RecompressMode ordinals (cmd/Optimise.java) = D4G_MODE_* */
    public static final int MODE_NONE = 0, MODE_CHEAP = 1, MODE_ZOPFLI = 2, MODE_ZOPFLI_EXTENSIVE = 3, MODE_ZOPFLI_VERY_EXTENSIVE = 4;
    /** D4G_ENC_*: which encoder family a GpuCompressor stands for */
    public static final int ENC_JVM = 0, ENC_JZLIB = 1;
    /** java.util.zip.Deflater strategy constants map 1:1: DEFAULT_STRATEGY 0, FILTERED 1, HUFFMAN_ONLY 2 */
    public static final int STRATEGY_DEFAULT = 0, STRATEGY_FILTERED = 1, STRATEGY_HUFFMAN_ONLY = 2;

    static {
        System.loadLibrary("deft4g_jni");
        // -Ddeft4g.devices=0,1,2,3,4,5,6,7: one context per GPU of the node (a JVM is one process); otherwise the one device
        final String devs = System.getProperty("deft4g.devices");
        final int rc;
        if (devs != null) {
            final String[] parts = devs.split(",");
            final int[] ids = new int[parts.length];
            for (int i = 0; i < parts.length; i++) {
                ids[i] = Integer.parseInt(parts[i].trim());
            }
            rc = initDevices(ids);
        } else {
            rc = init(Integer.getInteger("deft4g.device", 0));
        }
        if (rc != 0) {
            throw new UnsatisfiedLinkError("deft4g: no usable MI355X (d4g_init returned " + rc + "); there is no CPU fallback in the native library");
        }
    }

    private NativeDeft() {
    }

    private static native int init(int device);

    private static native int initDevices(int[] devices);

    /** contexts (GPUs) initialised */
    public static native int deviceCount();

    /** the calling thread's context from now on: a pool thread per GPU (CompressionUtil's pool) calls this once */
    public static native int setDevice(int context);

    /** optimiseStreams over every context: the list is split by size, one device batch per GPU, results in list order */
    public static native byte[][] optimiseStreamsSharded(byte[][] in, boolean mergeBlocks, long[] savedBits, int[] status);

    /** status[i]: 0 changed (result[i] holds the new bytes), 1 unchanged, -1 parse error (result[i] == null: keep the original array) */
    public static native byte[][] optimiseStreams(byte[][] in, boolean mergeBlocks, long[] savedBits, int[] status);

    public static native long sizeBitsFallback(byte[] in);

    /** SingleCompressor.compressSingle for every buffer */
    public static native byte[][] deflateStreams(byte[][] raw, int encoder, int strategy) throws java.io.IOException;

    /** MultiCafeUndZopfliCompressor / MultiJZopfliCompressor.compressWithOptions for every buffer (splitting: 0 FIRST, 1 LAST, 2 NONE) */
    public static native byte[][] zopfliStreams(byte[][] raw, int iterations, int splitting, int maxBlocks, long masterBlock) throws java.io.IOException;

    /** CompressionUtil.compress(uncompressedData, threaded) for every buffer */
    public static native byte[][] compress(byte[][] raw, int mode, int iter, boolean mergeBlocks) throws java.io.IOException;

    /** CMDUtil.optimise's per-stream work; result[i] == null: keep the input stream */
    public static native byte[][] recompressStreams(byte[][] in, int mode, int iter, boolean mergeBlocks, long[] savedBits, long[] recompressSaved,
            int[] status) throws java.io.IOException;

    /** Deft.optimiseDeflateStream(byte[], boolean): same contract, including "returns the SAME array when nothing was saved" */
    public static byte[] optimiseDeflateStream(byte[] original, boolean mergeBlocks) {
        final int[] status = new int[1];
        final byte[][] out = optimiseStreams(new byte[][] { original }, mergeBlocks, new long[1], status);
        return (out != null) && (status[0] == 0) ? out[0] : original;
    }
}
