// tests/jni_stub/jni.h — SYNTAX-CHECK ONLY.  The build container has no JDK; this declares just enough of the JNI C++ surface
// for `g++ -fsyntax-only jni/fspann_jni.cpp` to type-check the generated shim (tests/test_abi.py).  It is never linked,
// shipped or used to build anything; a real build uses $JAVA_HOME/include/jni.h (jni/Makefile).
#pragma once
#include <cstdint>
typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef jint jsize;
typedef unsigned char jboolean;
class _jobject {};
typedef _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;
typedef jarray jobjectArray;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
struct JNIEnv {
    void* GetDirectBufferAddress(jobject);
    jobject NewDirectByteBuffer(void*, jlong);
    jsize GetArrayLength(jarray);
    void GetIntArrayRegion(jintArray, jsize, jsize, jint*);
    void GetLongArrayRegion(jlongArray, jsize, jsize, jlong*);
    void SetIntArrayRegion(jintArray, jsize, jsize, const jint*);
    void SetLongArrayRegion(jlongArray, jsize, jsize, const jlong*);
    void SetDoubleArrayRegion(jdoubleArray, jsize, jsize, const jdouble*);
    const char* GetStringUTFChars(jstring, jboolean*);
    void ReleaseStringUTFChars(jstring, const char*);
    jstring NewStringUTF(const char*);
    void SetObjectArrayElement(jobjectArray, jsize, jobject);
};
