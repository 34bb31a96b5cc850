/* deft4g_jni.c — the thin JNI layer between deft4j's Java host code and libdeft4g.so (include/deft4g.h).
 *
 * Binds com.github.NeRdTheNed.deft4j.NativeDeft (java/com/github/NeRdTheNed/deft4j/NativeDeft.java).  Each function
 * copies Java byte[]s in, makes ONE C-ABI call for the whole list (one device batch) and copies the results out; no
 * logic lives here.  The build image has no JDK (no jni.h), so this file is compiled only where one exists:
 * `make -C jni` checks $JAVA_HOME/include/jni.h and skips otherwise.
 *
 * Reference seams replaced (paths under deft4j's source tree):
 *   Deft.optimiseDeflateStream / getSizeBitsFallback   deft4j-base/.../Deft.java:21-34,48-54
 *   DeflateFilesContainer.optimise(List, boolean)      deft4j-container/.../DeflateFilesContainer.java:18-43
 *   SingleCompressor.compressSingle                    deft4j-compress/.../util/compression/SingleCompressor.java:6
 *   CompressionUtil.compress                           deft4j-compress/.../util/compression/CompressionUtil.java:106-182
 *   CMDUtil.optimise's recompress loop                 deft4j-cmd/.../cmd/CMDUtil.java:76-105
 */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "deft4g.h"

#define JFN(name) Java_com_github_NeRdTheNed_deft4j_NativeDeft_##name

typedef struct {
    jsize n;
    jbyteArray* arr;
    const uint8_t** ptr;
    size_t* len;
} in_list;

static int pin(JNIEnv* e, jobjectArray in, in_list* L) {
    L->n = (*e)->GetArrayLength(e, in);
    /* one local reference per element stays alive until unpin(): a batch of a thousand members is far beyond the 16 the JVM
     * guarantees */
    if ((*e)->EnsureLocalCapacity(e, L->n + 16) != 0) return -1;
    L->arr = calloc((size_t)L->n + 1, sizeof *L->arr);
    L->ptr = calloc((size_t)L->n + 1, sizeof *L->ptr);
    L->len = calloc((size_t)L->n + 1, sizeof *L->len);
    if (!L->arr || !L->ptr || !L->len) return -1;
    for (jsize i = 0; i < L->n; i++) {
        L->arr[i] = (jbyteArray)(*e)->GetObjectArrayElement(e, in, i);
        if (!L->arr[i]) return -1;                                   /* a null element: NullPointerException territory in the reference too */
        L->len[i] = (size_t)(*e)->GetArrayLength(e, L->arr[i]);
        L->ptr[i] = (const uint8_t*)(*e)->GetByteArrayElements(e, L->arr[i], NULL);
        if (!L->ptr[i]) return -1;
    }
    return 0;
}
static void unpin(JNIEnv* e, in_list* L) {
    for (jsize i = 0; i < L->n; i++) {
        if (L->ptr && L->ptr[i]) (*e)->ReleaseByteArrayElements(e, L->arr[i], (jbyte*)L->ptr[i], JNI_ABORT);
        if (L->arr && L->arr[i]) (*e)->DeleteLocalRef(e, L->arr[i]);
    }
    free(L->arr); free((void*)L->ptr); free(L->len);
}
static void throw_io(JNIEnv* e, const char* what) {
    char msg[512];
    snprintf(msg, sizeof msg, "%s: %s", what, d4g_last_error());
    (*e)->ThrowNew(e, (*e)->FindClass(e, "java/io/IOException"), msg);
}
/* out[i] (NULL entries stay null) -> byte[][]; frees the library buffers */
static jobjectArray to_java(JNIEnv* e, jsize n, uint8_t** out, const size_t* olen) {
    jclass bytes = (*e)->FindClass(e, "[B");
    jobjectArray res = bytes ? (*e)->NewObjectArray(e, n, bytes, NULL) : NULL;
    for (jsize i = 0; i < n; i++) {
        if (out[i]) {
            /* (after a failed allocation an OutOfMemoryError is pending: no more JNI calls that may not run with one, only frees) */
            if (res && !(*e)->ExceptionCheck(e)) {
                jbyteArray o = (*e)->NewByteArray(e, (jsize)olen[i]);
                if (o) {
                    (*e)->SetByteArrayRegion(e, o, 0, (jsize)olen[i], (const jbyte*)out[i]);
                    (*e)->SetObjectArrayElement(e, res, i, o);
                    (*e)->DeleteLocalRef(e, o);
                }
            }
            d4g_free(out[i]);
        }
    }
    return (*e)->ExceptionCheck(e) ? NULL : res;
}

JNIEXPORT jint JNICALL JFN(init)(JNIEnv* e, jclass c, jint device) {
    (void)e; (void)c;
    return d4g_init(device);
}

/* int initDevices(int[] devices): context k = devices[k] (one JVM driving the GPUs of a node) */
JNIEXPORT jint JNICALL JFN(initDevices)(JNIEnv* e, jclass c, jintArray devices) {
    (void)c;
    jsize n = (*e)->GetArrayLength(e, devices);
    jint* d = (*e)->GetIntArrayElements(e, devices, NULL);
    if (!d) return D4G_ERR_RUNTIME;
    int rc = d4g_init_devices((int)n, (const int*)d);
    (*e)->ReleaseIntArrayElements(e, devices, d, JNI_ABORT);
    return rc;
}
JNIEXPORT jint JNICALL JFN(deviceCount)(JNIEnv* e, jclass c) {
    (void)e; (void)c;
    return d4g_device_count();
}
/* the calling Java thread's context from now on (a pool thread per GPU: CompressionUtil.java:111-117) */
JNIEXPORT jint JNICALL JFN(setDevice)(JNIEnv* e, jclass c, jint context) {
    (void)e; (void)c;
    return d4g_set_device(context);
}

/* byte[][] optimiseStreams(byte[][] in, boolean mergeBlocks, long[] savedBits, int[] status): entry i of the result is
 * null unless status[i] == 0 (changed) — the Java caller then returns its ORIGINAL array (Deft.java:33).  A library
 * failure reports every stream as "keep the original" (status 1): Deft.optimiseDeflateStream never throws. */
JNIEXPORT jobjectArray JNICALL JFN(optimiseStreams)(JNIEnv* e, jclass c, jobjectArray in, jboolean merge, jlongArray savedOut, jintArray statusOut) {
    (void)c;
    in_list L = {0};
    jobjectArray res = NULL;
    if (pin(e, in, &L) == 0) {
        uint8_t** out = calloc((size_t)L.n + 1, sizeof *out);
        size_t* olen = calloc((size_t)L.n + 1, sizeof *olen);
        int64_t* saved = calloc((size_t)L.n + 1, sizeof *saved);
        int32_t* status = calloc((size_t)L.n + 1, sizeof *status);
        if (d4g_optimise_streams((size_t)L.n, L.ptr, L.len, merge ? 1 : 0, out, olen, saved, status) != D4G_OK)
            for (jsize i = 0; i < L.n; i++) { status[i] = D4G_STREAM_UNCHANGED; saved[i] = 0; }   /* out[] is all NULL on failure */
        res = to_java(e, L.n, out, olen);
        if (res && (*e)->GetArrayLength(e, savedOut) >= L.n) (*e)->SetLongArrayRegion(e, savedOut, 0, L.n, (const jlong*)saved);
        if (res && (*e)->GetArrayLength(e, statusOut) >= L.n) (*e)->SetIntArrayRegion(e, statusOut, 0, L.n, (const jint*)status);
        free(out); free(olen); free(saved); free(status);
    }
    unpin(e, &L);
    return res;
}

/* the same over every initialised context: the list is partitioned by size, one device batch per context, outputs gathered
 * in list order (DeflateFilesContainer.optimise, K/DeflateFilesContainer.java:18-43) */
JNIEXPORT jobjectArray JNICALL JFN(optimiseStreamsSharded)(JNIEnv* e, jclass c, jobjectArray in, jboolean merge, jlongArray savedOut, jintArray statusOut) {
    (void)c;
    in_list L = {0};
    jobjectArray res = NULL;
    if (pin(e, in, &L) == 0) {
        uint8_t** out = calloc((size_t)L.n + 1, sizeof *out);
        size_t* olen = calloc((size_t)L.n + 1, sizeof *olen);
        int64_t* saved = calloc((size_t)L.n + 1, sizeof *saved);
        int32_t* status = calloc((size_t)L.n + 1, sizeof *status);
        if (d4g_optimise_streams_sharded((size_t)L.n, L.ptr, L.len, merge ? 1 : 0, out, olen, saved, status) != D4G_OK)
            for (jsize i = 0; i < L.n; i++) { status[i] = D4G_STREAM_UNCHANGED; saved[i] = 0; }
        res = to_java(e, L.n, out, olen);
        if (res && (*e)->GetArrayLength(e, savedOut) >= L.n) (*e)->SetLongArrayRegion(e, savedOut, 0, L.n, (const jlong*)saved);
        if (res && (*e)->GetArrayLength(e, statusOut) >= L.n) (*e)->SetIntArrayRegion(e, statusOut, 0, L.n, (const jint*)status);
        free(out); free(olen); free(saved); free(status);
    }
    unpin(e, &L);
    return res;
}

JNIEXPORT jlong JNICALL JFN(sizeBitsFallback)(JNIEnv* e, jclass c, jbyteArray a) {
    (void)c;
    jsize n = (*e)->GetArrayLength(e, a);
    jbyte* p = (*e)->GetByteArrayElements(e, a, NULL);
    int64_t bits = (int64_t)n * 8;                                   /* Deft.java:48-54: length * 8 on any failure */
    if (p) {
        d4g_size_bits_fallback((const uint8_t*)p, (size_t)n, &bits);
        (*e)->ReleaseByteArrayElements(e, a, p, JNI_ABORT);
    }
    return bits;
}

/* byte[][] deflateStreams(byte[][] raw, int encoder, int strategy): SingleCompressor.compressSingle for every buffer */
JNIEXPORT jobjectArray JNICALL JFN(deflateStreams)(JNIEnv* e, jclass c, jobjectArray raw, jint encoder, jint strategy) {
    (void)c;
    in_list L = {0};
    jobjectArray res = NULL;
    if (pin(e, raw, &L) == 0) {
        uint8_t** out = calloc((size_t)L.n + 1, sizeof *out);
        size_t* olen = calloc((size_t)L.n + 1, sizeof *olen);
        if (d4g_deflate_streams((size_t)L.n, L.ptr, L.len, encoder, strategy, out, olen) == D4G_OK) res = to_java(e, L.n, out, olen);
        else throw_io(e, "d4g_deflate_streams");
        free(out); free(olen);
    }
    unpin(e, &L);
    return res;
}

/* byte[][] zopfliStreams(byte[][] raw, int iterations, int splitting, int maxBlocks, long masterBlock):
 * MultiCafeUndZopfliCompressor / MultiJZopfliCompressor.compressWithOptions for every buffer */
JNIEXPORT jobjectArray JNICALL JFN(zopfliStreams)(JNIEnv* e, jclass c, jobjectArray raw, jint iterations, jint splitting, jint maxBlocks, jlong masterBlock) {
    (void)c;
    in_list L = {0};
    jobjectArray res = NULL;
    if (pin(e, raw, &L) == 0) {
        uint8_t** out = calloc((size_t)L.n + 1, sizeof *out);
        size_t* olen = calloc((size_t)L.n + 1, sizeof *olen);
        if (d4g_zopfli_streams((size_t)L.n, L.ptr, L.len, iterations, splitting, maxBlocks, (size_t)masterBlock, out, olen) == D4G_OK) res = to_java(e, L.n, out, olen);
        else throw_io(e, "d4g_zopfli_streams");
        free(out); free(olen);
    }
    unpin(e, &L);
    return res;
}

/* byte[][] compress(byte[][] raw, int mode, int iter, boolean mergeBlocks): CompressionUtil.compress for every buffer;
 * throws IOException("Unable to compress data") like CompressionUtil.java:177-179 */
JNIEXPORT jobjectArray JNICALL JFN(compress)(JNIEnv* e, jclass c, jobjectArray raw, jint mode, jint iter, jboolean merge) {
    (void)c;
    in_list L = {0};
    jobjectArray res = NULL;
    if (pin(e, raw, &L) == 0) {
        uint8_t** out = calloc((size_t)L.n + 1, sizeof *out);
        size_t* olen = calloc((size_t)L.n + 1, sizeof *olen);
        if (d4g_compress((size_t)L.n, L.ptr, L.len, mode, iter, merge ? 1 : 0, out, olen, NULL) == D4G_OK) res = to_java(e, L.n, out, olen);
        else throw_io(e, "Unable to compress data");
        free(out); free(olen);
    }
    unpin(e, &L);
    return res;
}

/* byte[][] recompressStreams(byte[][] in, int mode, int iter, boolean mergeBlocks, long[] savedBits, long[] recompressSaved,
 * int[] status): CMDUtil.optimise's per-stream work for a container's streams; null entries keep the original stream */
JNIEXPORT jobjectArray JNICALL JFN(recompressStreams)(JNIEnv* e, jclass c, jobjectArray in, jint mode, jint iter, jboolean merge, jlongArray savedOut,
                                                      jlongArray recompOut, jintArray statusOut) {
    (void)c;
    in_list L = {0};
    jobjectArray res = NULL;
    if (pin(e, in, &L) == 0) {
        uint8_t** out = calloc((size_t)L.n + 1, sizeof *out);
        size_t* olen = calloc((size_t)L.n + 1, sizeof *olen);
        int64_t* saved = calloc((size_t)L.n + 1, sizeof *saved);
        int64_t* rsaved = calloc((size_t)L.n + 1, sizeof *rsaved);
        int32_t* status = calloc((size_t)L.n + 1, sizeof *status);
        if (d4g_recompress_streams((size_t)L.n, L.ptr, L.len, mode, iter, merge ? 1 : 0, out, olen, saved, rsaved, status) == D4G_OK) {
            res = to_java(e, L.n, out, olen);
            (*e)->SetLongArrayRegion(e, savedOut, 0, L.n, (const jlong*)saved);
            (*e)->SetLongArrayRegion(e, recompOut, 0, L.n, (const jlong*)rsaved);
            (*e)->SetIntArrayRegion(e, statusOut, 0, L.n, (const jint*)status);
        } else throw_io(e, "d4g_recompress_streams");
        free(out); free(olen); free(saved); free(rsaved); free(status);
    }
    unpin(e, &L);
    return res;
}
