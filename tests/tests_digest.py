import hashlib

import numpy as np


def index_digest(ix):
    h = hashlib.sha256()
    for k in ("min_key", "max_key", "rep", "id_off", "ids"):
        h.update(np.ascontiguousarray(ix[k]).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()
