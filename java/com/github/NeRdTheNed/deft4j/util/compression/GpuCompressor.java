package com.github.NeRdTheNed.deft4j.util.compression;

import java.io.IOException;

import com.github.NeRdTheNed.deft4j.NativeDeft;

/**
 * A SingleCompressor (util/compression/SingleCompressor.java:6) backed by the MI355X LZ77 kernels: byte-identical to
 * JavaCompressor(strategy) for encoder ENC_JVM (zlib level 9; pinned by the asyoulik fixture), the jzlib restatement for
 * ENC_JZLIB.  CompressionUtil.getCompressors (CompressionUtil.java:44-78) would add
 *     new GpuCompressor(NativeDeft.ENC_JVM, Deflater.DEFAULT_STRATEGY) ...
 * in place of the JavaCompressor / JZLibCompressor entries, keeping the list order.
 */
public class GpuCompressor implements SingleCompressor {
    private final int encoder;
    private final int strategy;

    public GpuCompressor(int encoder, int strategy) {
        this.encoder = encoder;
        this.strategy = strategy;
    }

    @Override
    public byte[] compressSingle(byte[] uncompressedData) throws IOException {
        return NativeDeft.deflateStreams(new byte[][] { uncompressedData }, encoder, strategy)[0];
    }

    @Override
    public String getName() {
        return encoder == NativeDeft.ENC_JVM ? "GPU-zlib" : "GPU-jzlib";
    }
}
