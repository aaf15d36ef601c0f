"""Test collaborator: the reference's HOST side (unchanged subsystems) restated with OpenSSL.

AES-256-GCM exactly as crypto/AesGcmCryptoService.java:55-112,126-166,240-277: 12-byte random IV, 128-bit tag
appended to the ciphertext (javax.crypto doFinal layout), AAD "id:%s|v:%d|d:%d" (common/EncryptedPoint.java:80-83),
payload = 8*dim bytes big-endian fp64; queries are encrypted without AAD (:169-204).  Key versions:
K_v = HMAC-SHA256(K_M, int32_be(v)) (keymanagement/KeyManager.java:221-237); rotate / re-encrypt / retire as
KeyRotationServiceImpl.java:215-334.  Only used by tests: crypto stays on the host and out of scope.
"""
import ctypes as C
import ctypes.util
import hashlib
import hmac
import os
import struct
import threading

import numpy as np

_lc = C.CDLL(ctypes.util.find_library("crypto"))
_lc.EVP_CIPHER_CTX_new.restype = C.c_void_p
_lc.EVP_aes_256_gcm.restype = C.c_void_p
for _n in ("EVP_EncryptInit_ex", "EVP_DecryptInit_ex"):
    getattr(_lc, _n).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p]
for _n in ("EVP_EncryptUpdate", "EVP_DecryptUpdate"):
    getattr(_lc, _n).argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.c_int]
for _n in ("EVP_EncryptFinal_ex", "EVP_DecryptFinal_ex"):
    getattr(_lc, _n).argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
_lc.EVP_CIPHER_CTX_ctrl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
_lc.EVP_CIPHER_CTX_free.argtypes = [C.c_void_p]
_GET_TAG, _SET_TAG, _SET_IVLEN = 0x10, 0x11, 0x9


class AuthError(Exception):
    pass


def gcm_encrypt(key: bytes, iv: bytes, pt: bytes, aad: bytes = b"") -> bytes:
    ctx = _lc.EVP_CIPHER_CTX_new()
    try:
        n = C.c_int(0)
        assert _lc.EVP_EncryptInit_ex(ctx, _lc.EVP_aes_256_gcm(), None, None, None) == 1
        assert _lc.EVP_CIPHER_CTX_ctrl(ctx, _SET_IVLEN, len(iv), None) == 1
        assert _lc.EVP_EncryptInit_ex(ctx, None, None, key, iv) == 1
        if aad:
            assert _lc.EVP_EncryptUpdate(ctx, None, C.byref(n), aad, len(aad)) == 1
        out = C.create_string_buffer(len(pt) + 16)
        assert _lc.EVP_EncryptUpdate(ctx, out, C.byref(n), pt, len(pt)) == 1
        ln = n.value
        assert _lc.EVP_EncryptFinal_ex(ctx, C.cast(C.addressof(out) + ln, C.c_char_p), C.byref(n)) == 1
        tag = C.create_string_buffer(16)
        assert _lc.EVP_CIPHER_CTX_ctrl(ctx, _GET_TAG, 16, tag) == 1
        return out.raw[:ln] + tag.raw
    finally:
        _lc.EVP_CIPHER_CTX_free(ctx)


def gcm_decrypt(key: bytes, iv: bytes, ct_tag: bytes, aad: bytes = b"") -> bytes:
    ct, tag = ct_tag[:-16], ct_tag[-16:]
    ctx = _lc.EVP_CIPHER_CTX_new()
    try:
        n = C.c_int(0)
        assert _lc.EVP_DecryptInit_ex(ctx, _lc.EVP_aes_256_gcm(), None, None, None) == 1
        assert _lc.EVP_CIPHER_CTX_ctrl(ctx, _SET_IVLEN, len(iv), None) == 1
        assert _lc.EVP_DecryptInit_ex(ctx, None, None, key, iv) == 1
        if aad:
            assert _lc.EVP_DecryptUpdate(ctx, None, C.byref(n), aad, len(aad)) == 1
        out = C.create_string_buffer(max(len(ct), 1))
        assert _lc.EVP_DecryptUpdate(ctx, out, C.byref(n), ct, len(ct)) == 1
        ln = n.value
        assert _lc.EVP_CIPHER_CTX_ctrl(ctx, _SET_TAG, 16, C.create_string_buffer(tag, 16)) == 1
        if _lc.EVP_DecryptFinal_ex(ctx, C.cast(C.addressof(out) + ln, C.c_char_p), C.byref(n)) != 1:
            raise AuthError("GCM tag mismatch")
        return out.raw[:ln]
    finally:
        _lc.EVP_CIPHER_CTX_free(ctx)


class AesGcmHost:
    """CryptoService + KeyLifeCycleService + metadata store with real AES-256-GCM."""

    def __init__(self, master=None):
        from fspann_amd import operators as ops
        self.ops = ops
        self.master = master or os.urandom(32)
        self.version = 1
        self.retired = set()
        self.points = {}
        self.deleted = set()
        self.lock = threading.Lock()
        self.decrypt_count = 0

    # --- keys (KeyManager.deriveKey) ---------------------------------------------------
    def _key(self, v):
        if v in self.retired:
            raise KeyError(f"key version {v} retired")
        return hmac.new(self.master, struct.pack(">i", v), hashlib.sha256).digest()

    def getCurrentVersion(self):
        return self.ops.KeyVersion(self.version, self._key(self.version))

    def getVersion(self, v):
        return self.ops.KeyVersion(v, self._key(v))

    def rotateKeyOnly(self):                       # KeyRotationServiceImpl.java:292-305
        with self.lock:
            self.version += 1
            return self.version

    def retire(self, v):                           # KeyManager.java:274-317
        self.retired.add(v)

    # --- crypto ------------------------------------------------------------------------
    @staticmethod
    def _aad(id, version, dim):
        return ("id:%s|v:%d|d:%d" % (id, version, dim)).encode()

    def encrypt(self, id, vector, kv=None):
        kv = kv or self.getCurrentVersion()
        iv = os.urandom(12)
        pt = np.asarray(vector, dtype=">f8").tobytes()
        ct = gcm_encrypt(kv.key, iv, pt, self._aad(id, kv.version, len(vector)))
        return self.ops.EncryptedPoint(id, kv.version, iv, ct, len(vector))

    def decryptFromPoint(self, ep, key):
        self.decrypt_count += 1
        pt = gcm_decrypt(key, ep.iv, ep.ciphertext, self._aad(ep.id, ep.version, ep.dim))
        return np.frombuffer(pt, dtype=">f8").astype(np.float64)

    def encryptQuery(self, vec, key, iv):
        return gcm_encrypt(key, iv, np.asarray(vec, dtype=">f8").tobytes())

    def decryptQuery(self, ct, iv, key):
        return np.frombuffer(gcm_decrypt(key, iv, ct), dtype=">f8").astype(np.float64)

    # --- metadata ----------------------------------------------------------------------
    def saveEncryptedPoint(self, ep):
        with self.lock:
            self.points[ep.id] = ep

    def loadEncryptedPoint(self, id):
        with self.lock:
            return self.points.get(id)

    def isDeleted(self, id):
        return id in self.deleted

    # --- Migrate (KeyRotationServiceImpl.reencryptTouched :215-289) ---------------------------
    def reencrypt(self, ids, target_version):
        n = 0
        for id in ids:
            ep = self.loadEncryptedPoint(id)
            if ep is None or ep.version >= target_version:
                continue
            v = self.decryptFromPoint(ep, self._key(ep.version))
            self.saveEncryptedPoint(self.encrypt(id, v, self.getVersion(target_version)))
            n += 1
        return n
