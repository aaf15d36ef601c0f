#!/usr/bin/env python3
"""Generates jni/fspann_jni.cpp and java/com/fspann/gpu/FspannNative.java from include/fspann.h: ONE native method per C
entry point, pure marshalling.  Run after the header changes (tests/test_abi.py checks that every header symbol is bound).

Mapping: opaque handles (fspann_ctx*, fspann_comm*, fspann_pointstore*, fspann_pipeline*) and device pointers (parameters
whose name ends in _dev) travel as `long`; host arrays as DIRECT java.nio.ByteBuffer in native byte order (null -> NULL);
scalar outputs as one-element arrays; fspann_cfg as int[12]; fspann_tick as long[29] in field order.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "fspann.h")
HANDLES = {"fspann_ctx": "ctx", "fspann_comm": "comm", "fspann_pointstore": "ps", "fspann_pipeline": "pipe"}
# scalar outputs (function, parameter) -> kind
OUT = {
    ("fspann_index_dims", "n_parts"): "long", ("fspann_index_dims", "n_ids"): "long",
    ("fspann_last_route_info", "lazy"): "int", ("fspann_last_route_info", "overflowed"): "int",
    ("fspann_unmodelled_queries", "total"): "long",
    ("fspann_refine_timing_end", "launches"): "int", ("fspann_refine_timing_end", "total_ms"): "double",
    ("fspann_store_dev_ptr", "dtype"): "int",
    ("fspann_pointstore_rotate", "new_version"): "int", ("fspann_pointstore_reencrypt", "reencrypted"): "long",
    ("fspann_pointstore_get_record", "version"): "int",
    ("fspann_pointstore_stats", "opened"): "long", ("fspann_pointstore_stats", "failed"): "long",
    ("fspann_pipeline_submit", "ticket"): "long", ("fspann_pipeline_collect", "ticket"): "long", ("fspann_pipeline_collect", "nq"): "long",
    ("fspann_pipeline_stats", "route_ms"): "double", ("fspann_pipeline_stats", "decrypt_ms"): "double",
    ("fspann_pipeline_stats", "refine_ms"): "double", ("fspann_pipeline_stats", "batches"): "long",
    ("fspann_comm_info", "world"): "int", ("fspann_comm_info", "rank"): "int",
    ("fspann_hbm_read_peak", "gb_per_s"): "double",
    ("fspann_hbm_read_window", "gb_per_s"): "double",
}
TICK_FIELDS = ["nq_encode", "enc_q_dev", "enc_dtype", "pad0", "enc_codes_dev", "enc_bad_dev", "nq_route", "route_codes_dev",
               "route_probe_override", "route_limit", "route_ids_dev", "route_count_dev", "route_handover_dev", "nq_refine", "ref_q_dev",
               "ref_q_dtype", "ref_cand_dtype", "ref_cand_dev", "ref_B", "ref_ids_dev", "ref_count_dev", "ref_codes_dev", "ref_handover_dev",
               "ref_probe_override", "k", "out_ids_dev", "out_dist_dev", "out_count_dev", "scored_dev"]
TICK_PTR = {f for f in TICK_FIELDS if f.endswith("_dev")}


def prototypes():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"typedef struct \w+ \{.*?\} \w+;", " ", src, flags=re.S)
    out = []
    for m in re.finditer(r"(?m)^((?:const\s+)?[\w\s\*]+?)\b(fspann_\w+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, params = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        ps = []
        if params and params != "void":
            for p in params.split(","):
                p = p.strip()
                mm = re.match(r"(.*?)(\w+)$", p)
                ps.append((" ".join(mm.group(1).split()), mm.group(2)))
        out.append((ret, name, ps))
    return out


RENAME = {"fspann_finalize": "finalizeIndex"}        # (Object.finalize() would be shadowed by a static overload)


def camel(name):
    if name in RENAME:
        return RENAME[name]
    parts = name[len("fspann_"):].split("_")
    return parts[0] + "".join(x.capitalize() for x in parts[1:])


def classify(fn, ctype, pname):
    t = ctype.replace("const ", "").strip()
    base = t.rstrip("* ").strip()
    stars = t.count("*")
    if (fn, pname) in OUT:
        return "out_" + OUT[(fn, pname)]
    if base in HANDLES:
        return "handle" if stars == 1 else "out_handle"
    if base == "fspann_cfg":
        return "cfg"
    if base == "fspann_tick":
        return "tick"
    if base == "char":
        return "string" if stars == 1 else "out_string"
    if stars == 2 and base == "void":
        return "out_ptr"
    if stars >= 1:
        return "devptr" if pname.endswith("_dev") else "buffer"
    if base in ("int", "int32_t"):
        return "int"
    if base in ("int64_t", "size_t", "uint64_t"):
        return "long"
    raise SystemExit(f"gen_jni: cannot map {fn}({ctype} {pname})")


JAVA_T = {"handle": "long", "out_handle": "long[]", "cfg": "int[]", "tick": "long[]", "string": "String", "out_string": "String[]", "out_ptr": "long[]",
          "devptr": "long", "buffer": "ByteBuffer", "int": "int", "long": "long", "out_int": "int[]", "out_long": "long[]", "out_double": "double[]"}
JNI_T = {"handle": "jlong", "out_handle": "jlongArray", "cfg": "jintArray", "tick": "jlongArray", "string": "jstring", "out_string": "jobjectArray",
         "out_ptr": "jlongArray", "devptr": "jlong", "buffer": "jobject", "int": "jint", "long": "jlong", "out_int": "jintArray",
         "out_long": "jlongArray", "out_double": "jdoubleArray"}


def main():
    protos = prototypes()
    cpp = ['// GENERATED by tools/gen_jni.py from include/fspann.h — do not edit by hand.',
           '// JNI shim between com.fspann.gpu.FspannNative and the C ABI: pure marshalling, one native method per C entry point.',
           '// Every ByteBuffer must be DIRECT (the C side reads / writes it in place); null buffers become NULL pointers.',
           '// Build (needs a JDK; none exists in the build container, see INTEGRATION.md):',
           '//   g++ -O2 -fPIC -shared -I"$JAVA_HOME/include" -I"$JAVA_HOME/include/linux" -I../include -o libfspann_jni.so fspann_jni.cpp',
           '//       -L../fspann-query-system_amd -lfspann_hip          (jni/Makefile does this when JAVA_HOME is set)',
           '#include <jni.h>', '', '#include <cstdint>', '#include <cstring>', '', '#include "fspann.h"', '',
           'namespace {',
           'inline void* addr(JNIEnv* env, jobject buf) { return buf ? env->GetDirectBufferAddress(buf) : nullptr; }',
           'template <class T> inline T* H(jlong h) { return reinterpret_cast<T*>(static_cast<intptr_t>(h)); }',
           'inline jlong L(const void* p) { return static_cast<jlong>(reinterpret_cast<intptr_t>(p)); }',
           '}  // namespace', '', 'extern "C" {', '']
    java = ['// GENERATED by tools/gen_jni.py from include/fspann.h — do not edit by hand.', 'package com.fspann.gpu;', '',
            'import java.nio.ByteBuffer;', '', '/**',
            ' * JNI binding of libfspann_hip.so: one native method per entry point of include/fspann.h (same order, same argument meaning —',
            ' * the header is the documentation).  Opaque handles and device pointers (*_dev / *Dev parameters) are {@code long};',
            ' * host arrays are DIRECT ByteBuffers in native byte order, read and written in place; scalar outputs are one-element',
            ' * arrays; fspann_cfg is {@code int[12]} in field order; fspann_tick is {@code long[' + str(len(TICK_FIELDS)) + ']} in field order.',
            ' * Every int-returning method returns the C return code: {@link #check(int)} turns it into the exception class the',
            ' * reference itself would throw.  NOT compiled in the build container (no JDK there); jni/Makefile gates on JAVA_HOME.',
            ' */', 'public final class FspannNative {', '    static { System.loadLibrary("fspann_jni"); }   // links libfspann_hip.so', '',
            '    private FspannNative() {}', '', '    public static final int F32 = 0, F64 = 1;',
            '    public static final int OK = 0, E_STATE = -1, E_ARG = -2, E_NULL = -3, E_DEVICE = -4, E_NOMEM = -5, E_RANGE = -6;',
            '    /** fspann_tick field order for the long[] passed to tickDev. */',
            '    public static final String[] TICK_FIELDS = {' + ", ".join('"%s"' % f for f in TICK_FIELDS) + '};', '']
    names = []
    for ret, name, ps in protos:
        jn = camel(name)
        names.append(name)
        kinds = [(classify(name, t, p), t, p) for t, p in ps]
        rett = ret.replace("const ", "").strip()
        if rett == "int":
            jret, jniret = "int", "jint"
        elif rett == "void":
            jret, jniret = "void", "void"
        elif rett == "char*":
            jret, jniret = "String", "jstring"
        elif rett in ("size_t", "int64_t"):
            jret, jniret = "long", "jlong"
        elif rett == "void*" and name == "fspann_host_buffer":
            jret, jniret = "ByteBuffer", "jobject"      # the pinned block itself, as a direct buffer the adapter fills in place
        elif rett == "void*":
            jret, jniret = "long", "jlong"
        else:
            raise SystemExit(f"gen_jni: return type {ret} of {name}")
        jparams = ", ".join(f"{JAVA_T[k]} {re.sub(r'_(.)', lambda m: m.group(1).upper(), p)}" for k, t, p in kinds)
        java.append(f"    public static native {jret} {jn}({jparams});")
        sig = ", ".join(["JNIEnv* env", "jclass"] + [f"{JNI_T[k]} {p}" for k, t, p in kinds])
        cpp.append(f"JNIEXPORT {jniret} JNICALL Java_com_fspann_gpu_FspannNative_{jn}({sig}) {{")
        pre, args, post = [], [], []
        for k, t, p in kinds:
            base = t.replace("const ", "").rstrip("* ").strip()
            if k == "handle":
                args.append(f"H<{base}>({p})")
            elif k == "out_handle":
                pre.append(f"    {base}* {p}_v = nullptr;")
                args.append(f"&{p}_v")
                post.append(f"    if ({p}) {{ jlong v = L({p}_v); env->SetLongArrayRegion({p}, 0, 1, &v); }}")
            elif k == "cfg":
                pre.append(f"    jint {p}_i[12] = {{0}};")
                pre.append(f"    if ({p}) {{ jsize n_ = env->GetArrayLength({p}); env->GetIntArrayRegion({p}, 0, n_ < 12 ? n_ : 12, {p}_i); }}")
                pre.append(f"    fspann_cfg {p}_c = {{{', '.join(f'{p}_i[{i}]' for i in range(12))}}};")
                args.append(f"{p} ? &{p}_c : nullptr")
            elif k == "tick":
                n = len(TICK_FIELDS)
                pre.append(f"    jlong {p}_l[{n}] = {{0}};")
                pre.append(f"    if ({p}) {{ jsize n_ = env->GetArrayLength({p}); env->GetLongArrayRegion({p}, 0, n_ < {n} ? n_ : {n}, {p}_l); }}")
                pre.append(f"    fspann_tick {p}_s;")
                pre.append(f"    std::memset(&{p}_s, 0, sizeof({p}_s));")
                for i, f in enumerate(TICK_FIELDS):
                    if f in TICK_PTR:
                        pre.append(f"    {p}_s.{f} = reinterpret_cast<decltype({p}_s.{f})>(static_cast<intptr_t>({p}_l[{i}]));")
                    else:
                        pre.append(f"    {p}_s.{f} = static_cast<decltype({p}_s.{f})>({p}_l[{i}]);")
                args.append(f"{p} ? &{p}_s : nullptr")
            elif k == "string":
                pre.append(f"    const char* {p}_s = {p} ? env->GetStringUTFChars({p}, nullptr) : nullptr;")
                args.append(f"{p}_s")
                post.append(f"    if ({p}_s) env->ReleaseStringUTFChars({p}, {p}_s);")
            elif k == "out_string":
                pre.append(f"    const char* {p}_v = nullptr;")
                args.append(f"&{p}_v")
                post.append(f"    if ({p} && {p}_v) env->SetObjectArrayElement({p}, 0, env->NewStringUTF({p}_v));")
            elif k == "out_ptr":
                pre.append(f"    void* {p}_v = nullptr;")
                args.append(f"&{p}_v")
                post.append(f"    if ({p}) {{ jlong v = L({p}_v); env->SetLongArrayRegion({p}, 0, 1, &v); }}")
            elif k == "devptr":
                args.append(f"reinterpret_cast<{t}>(static_cast<intptr_t>({p}))")
            elif k == "buffer":
                args.append(f"static_cast<{t}>(addr(env, {p}))")
            elif k == "int":
                args.append(f"static_cast<{base}>({p})")
            elif k == "long":
                args.append(f"static_cast<{base}>({p})")
            elif k.startswith("out_"):
                ct = {"out_int": ("jint", "Int", base), "out_long": ("jlong", "Long", base), "out_double": ("jdouble", "Double", "double")}[k]
                pre.append(f"    {ct[2]} {p}_v = 0;")
                args.append(f"{p} ? &{p}_v : nullptr")
                post.append(f"    if ({p}) {{ {ct[0]} v = static_cast<{ct[0]}>({p}_v); env->Set{ct[1]}ArrayRegion({p}, 0, 1, &v); }}")
        call = f"{name}({', '.join(args)})"
        cpp += pre
        if jniret == "void":
            cpp.append(f"    {call};")
            cpp += post
        elif jniret == "jstring":
            cpp.append(f"    const char* r_ = {call};")
            cpp += post
            cpp.append("    return env->NewStringUTF(r_ ? r_ : \"\");")
        elif jniret == "jobject":
            cpp.append(f"    void* r_ = {call};")
            cpp += post
            cpp.append("    return r_ ? env->NewDirectByteBuffer(r_, static_cast<jlong>(bytes)) : nullptr;")
        elif rett == "void*":
            cpp.append(f"    jlong r_ = L({call});")
            cpp += post
            cpp.append("    return r_;")
        else:
            cpp.append(f"    {jniret} r_ = static_cast<{jniret}>({call});")
            cpp += post
            cpp.append("    return r_;")
        if jniret != "jobject" and (not kinds or all(k not in ("buffer", "string", "out_handle", "cfg", "tick", "out_string", "out_ptr", "out_int", "out_long", "out_double")
                            for k, _, _ in kinds)):
            cpp.insert(len(cpp) - (len(pre) + len(post) + (1 if jniret == "void" else 2)), "    (void)env;")
        cpp.append("}")
        cpp.append("")
    cpp.append('}  // extern "C"')
    java += ['', '    /** C return code -> the exception class the reference would throw (include/fspann.h). */',
             '    public static void check(int rc) {', '        if (rc == OK) return;', '        String msg = lastError();',
             '        switch (rc) {', '            case E_STATE: throw new IllegalStateException(msg);',
             '            case E_ARG: throw new IllegalArgumentException(msg);', '            case E_NULL: throw new NullPointerException(msg);',
             '            case E_NOMEM: throw new OutOfMemoryError(msg);',
             '            default: throw new RuntimeException("fspann(" + rc + "): " + msg);', '        }', '    }', '}', '']
    open(os.path.join(ROOT, "jni", "fspann_jni.cpp"), "w").write("\n".join(cpp) + "\n")
    open(os.path.join(ROOT, "java", "com", "fspann", "gpu", "FspannNative.java"), "w").write("\n".join(java))
    open(os.path.join(ROOT, "jni", "bound_symbols.txt"), "w").write("\n".join(names) + "\n")
    print(f"gen_jni: {len(names)} entry points bound")


if __name__ == "__main__":
    main()
