/* TEST INFRASTRUCTURE ONLY — a declaration-only jni.h, just enough for `gcc -fsyntax-only jni/ldbg_jni.c` in an image without a JDK
 * (tests/test_jni_glue.py).  Types and member names follow the JNI specification (JavaSE "Java Native Interface Specification",
 * chapter 4); nothing here is linked or shipped.  A real build uses $JAVA_HOME/include/jni.h. */
#ifndef LDBG_TEST_JNI_STUB_H
#define LDBG_TEST_JNI_STUB_H
#include <stdint.h>
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_ABORT 2
typedef int32_t jint; typedef int64_t jlong; typedef int8_t jbyte; typedef uint8_t jboolean; typedef jint jsize;
struct _jobject; typedef struct _jobject* jobject;
typedef jobject jclass; typedef jobject jstring; typedef jobject jarray; typedef jobject jthrowable;
typedef jarray jobjectArray; typedef jarray jbyteArray; typedef jarray jintArray; typedef jarray jlongArray;
struct JNINativeInterface_; typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv*, const char*);
    jint (*ThrowNew)(JNIEnv*, jclass, const char*);
    void (*ExceptionClear)(JNIEnv*);
    jstring (*NewStringUTF)(JNIEnv*, const char*);
    const char* (*GetStringUTFChars)(JNIEnv*, jstring, jboolean*);
    void (*ReleaseStringUTFChars)(JNIEnv*, jstring, const char*);
    jsize (*GetArrayLength)(JNIEnv*, jarray);
    jobject (*GetObjectArrayElement)(JNIEnv*, jobjectArray, jsize);
    void (*SetObjectArrayElement)(JNIEnv*, jobjectArray, jsize, jobject);
    jbyteArray (*NewByteArray)(JNIEnv*, jsize);
    jintArray (*NewIntArray)(JNIEnv*, jsize);
    jlongArray (*NewLongArray)(JNIEnv*, jsize);
    jbyte* (*GetByteArrayElements)(JNIEnv*, jbyteArray, jboolean*);
    jint* (*GetIntArrayElements)(JNIEnv*, jintArray, jboolean*);
    jlong* (*GetLongArrayElements)(JNIEnv*, jlongArray, jboolean*);
    void (*ReleaseByteArrayElements)(JNIEnv*, jbyteArray, jbyte*, jint);
    void (*ReleaseIntArrayElements)(JNIEnv*, jintArray, jint*, jint);
    void (*ReleaseLongArrayElements)(JNIEnv*, jlongArray, jlong*, jint);
    void (*GetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, jbyte*);
    void (*GetIntArrayRegion)(JNIEnv*, jintArray, jsize, jsize, jint*);
    void (*SetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, const jbyte*);
    void (*SetIntArrayRegion)(JNIEnv*, jintArray, jsize, jsize, const jint*);
    void (*SetLongArrayRegion)(JNIEnv*, jlongArray, jsize, jsize, const jlong*);
};
#endif
