"""CPU: the packed point store speaks the reference's AES-256-GCM record format (crypto/AesGcmCryptoService.java:55-112,
126-166,240-277; common/EncryptedPoint.java:80-83; keymanagement/KeyManager.java:221-237): records sealed by the C++ store are
opened by the Python restatement of the reference's crypto (tests/aesgcm_host.py) and vice versa; Rotate / Migrate / Retire
(keymanagement/KeyRotationServiceImpl.java:215-334) behave like the reference's."""
import hashlib
import hmac
import struct

import numpy as np
import pytest

import __graft_entry__ as graft
from aesgcm_host import AuthError, gcm_decrypt, gcm_encrypt


def _key(master, v):
    return hmac.new(master, struct.pack(">i", v), hashlib.sha256).digest()


def _aad(h, v, d):
    return ("id:%s|v:%d|d:%d" % (h, v, d)).encode()


@pytest.fixture(scope="module")
def hp():
    graft.load_package()
    from fspann_amd import hostpipe
    return hostpipe


def test_records_open_with_the_reference_crypto(hp):
    rng = np.random.default_rng(0)
    n, d = 300, 24
    X = rng.standard_normal((n, d)).astype(np.float32)
    master = bytes(range(32))
    with hp.PointStore(n, d, master) as ps:
        ps.encrypt(X, threads=3)
        for h in (0, 1, 57, n - 1):
            ver, iv, ct = ps.get_record(h)
            assert ver == 1 and len(iv) == 12 and len(ct) == 8 * d + 16
            pt = gcm_decrypt(_key(master, 1), iv, ct, _aad(h, 1, d))            # the reference's decryptFromPoint
            assert np.array_equal(np.frombuffer(pt, dtype=">f8"), X[h].astype(np.float64))
            with pytest.raises(AuthError):                                       # AAD binds id, version and dimension
                gcm_decrypt(_key(master, 1), iv, ct, _aad(h + 1, 1, d))
            with pytest.raises(AuthError):
                gcm_decrypt(_key(master, 2), iv, ct, _aad(h, 1, d))
        # and the other way round: a record sealed by the reference's encryptToPoint is opened by the store
        v64 = rng.standard_normal(d)
        iv = bytes(range(12))
        ct = gcm_encrypt(_key(master, 1), iv, v64.astype(">f8").tobytes(), _aad(5, 1, d))
        ps.put_record(5, 1, iv, ct)
        ids = np.array([[5, 7, 5]], np.int32)
        rows, oi, oc = ps.open_batch(ids, np.array([3], np.int32), dtype=np.float64, threads=2)
        assert oc[0] == 3 and np.array_equal(oi[0], [5, 7, 5])
        assert np.array_equal(rows[0, 0], v64) and np.array_equal(rows[0, 1], X[7].astype(np.float64))
        # fresh IV per record (SecureRandom): no two records share one
        assert len({ps.get_record(h)[1] for h in range(n)}) >= n - 1


def test_failed_loads_are_skipped_and_survivors_packed(hp):
    rng = np.random.default_rng(1)
    n, d, B = 64, 8, 6
    X = rng.standard_normal((n, d)).astype(np.float32)
    with hp.PointStore(n, d) as ps:
        ps.encrypt(X)
        ps.delete(3)                                          # loadPointIfActive() == null
        ver, iv, ct = ps.get_record(9)
        bad = bytearray(ct)
        bad[5] ^= 1
        ps.put_record(9, ver, iv, bytes(bad))                 # corrupted ciphertext: tag mismatch, swallowed (QSI:264-269)
        ids = np.array([[1, 3, 2, 9, 4, 70], [9, 9, 9, 0, -1, 5]], np.int32)     # 70 / -1: not in the store
        rows, oi, oc = ps.open_batch(ids, np.array([6, 4], np.int32))
        assert np.array_equal(oc, [3, 1])
        assert np.array_equal(oi[0], [1, 2, 4, -1, -1, -1]) and np.array_equal(oi[1], [0, -1, -1, -1, -1, -1])
        assert np.array_equal(rows[0, :3], X[[1, 2, 4]]) and np.array_equal(rows[1, 0], X[0])
        st = ps.stats()
        assert st["opened"] == 4 and st["failed"] == 6
        rows, oi, oc = ps.open_batch(ids, np.array([-2, 0], np.int32))     # negative counts (PENDING / unmodelled) = nothing
        assert np.array_equal(oc, [0, 0])


def test_rotate_migrate_retire(hp):
    rng = np.random.default_rng(2)
    n, d = 200, 16
    X = rng.standard_normal((n, d)).astype(np.float32)
    master = bytes(reversed(range(32)))
    with hp.PointStore(n, d, master) as ps:
        ps.encrypt(X)
        before = {h: ps.get_record(h) for h in range(n)}
        assert ps.rotate() == 2 and ps.version == 2                       # rotateKeyOnly: no record touched
        assert all(ps.get_record(h) == before[h] for h in (0, 10, n - 1))
        touched = np.arange(0, n, 2, dtype=np.int32)
        assert ps.reencrypt(touched, threads=3) == len(touched)            # reencryptTouched
        assert ps.reencrypt(touched, threads=2) == 0                       # already upgraded -> skipped
        for h in (0, 2, 198):
            ver, iv, ct = ps.get_record(h)
            assert ver == 2 and iv != before[h][1] and ct != before[h][2]  # fresh IV, new ciphertext
            pt = gcm_decrypt(_key(master, 2), iv, ct, _aad(h, 2, d))
            assert np.array_equal(np.frombuffer(pt, dtype=">f8"), X[h].astype(np.float64))
            with pytest.raises(AuthError):                                 # the old key no longer opens it (forward security)
                gcm_decrypt(_key(master, 1), iv, ct, _aad(h, 2, d))
        assert ps.get_record(1) == before[1]                               # untouched ids keep their version-1 record
        ids = np.arange(n, dtype=np.int32).reshape(1, n)
        rows, oi, oc = ps.open_batch(ids, np.array([n], np.int32))
        assert oc[0] == n and np.array_equal(rows[0], X)                   # both generations readable
        ps.retire(1)                                                       # retire K_1: version-1 records become unreadable
        rows, oi, oc = ps.open_batch(ids, np.array([n], np.int32))
        assert oc[0] == len(touched) and np.array_equal(oi[0, :oc[0]], touched) and np.array_equal(rows[0, :oc[0]], X[touched])
        ps.rotate()
        assert ps.reencrypt(np.arange(n, dtype=np.int32)) == len(touched)  # only what is still readable can migrate


def test_errors(hp, ):
    pkg = graft.load_package()
    with pytest.raises(pkg.FspannArgumentError):
        hp.PointStore(0, 8)
    with hp.PointStore(10, 4) as ps:
        with pytest.raises(pkg.FspannArgumentError):
            ps.encrypt(np.zeros((11, 4), np.float32))
        with pytest.raises(pkg.FspannArgumentError):
            ps.get_record(10)
