// =============================================================================
// fspann_oracle.cpp — CPU ORACLE for FSPANN TokenGen -> Route -> Refine.
//
// THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//   * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
//     load liboracle.so.  Nothing under fspann-query-system_amd/ links, imports
//     or calls it; the product path fails loudly when the HIP library is absent.
//
// PARITY STATUS: **parity unpinned**.
//   The reference (Mehran-Memon/fspann-query-system, pure Java 21) ships no
//   golden vectors / known-answer tests for Coding, Route or Refine (its tests
//   pin one bit-ordering property, see orc_quickcheck()), and no JVM exists in
//   the build container, so the reference itself cannot be run here.  This file
//   is a line-by-line restatement of the reference sources, plus literal models
//   of the JDK collections whose iteration order is part of the reference's
//   observable behaviour (java.util.HashMap, java.util.PriorityQueue,
//   java.util.SplittableRandom, String.hashCode, (int) casts, BitSet).
//   The JDK models are written from the JDK 21 specification; they are pinned
//   only by the micro known-answer tests in tests/test_oracle_java_semantics.py
//   (SplitMix64 published vectors, String.hashCode published values).
//   `Math.log`/`Math.cos` are NOT bit-reproducible outside a JVM, so GFunction
//   parameters generated here are self-consistent but never claimed to equal the
//   JVM's for the same seed; the product boundary imports alpha/r/omega.
//
// Reference files restated (all under /root/reference/fsp-anns-parent/):
//   idx = index/src/main/java/com/fspann/index/paper
//   qry = query/src/main/java/com/fspann/query
//   idx/Coding.java:136-161,184-241,250-301,342-361
//   idx/GFunctionRegistry.java:63-147,291-293
//   idx/GreedyPartitioner.java:37-130
//   idx/PartitionedIndexService.java:265-347,372-434,459-582,592-753,789-845,880-888
//   qry/service/QueryServiceImpl.java:101-352,364-372,407-413,444-466
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -shared -fPIC
//        (see oracle/Makefile).  -ffp-contract=off matters: Java never fuses
//        a*b+c, so neither may this file.
// =============================================================================
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#if defined(_OPENMP)
#include <omp.h>
#endif

namespace {

// -----------------------------------------------------------------------------
// Java primitive semantics
// -----------------------------------------------------------------------------

// java.util.SplittableRandom(long seed): SplitMix64, gamma = GOLDEN_GAMMA.
// nextLong() = mix64(seed += gamma); nextDouble() = (nextLong() >>> 11) * 2^-53.
struct SplittableRandom {
    uint64_t seed;
    explicit SplittableRandom(int64_t s) : seed(static_cast<uint64_t>(s)) {}
    uint64_t nextLong() {
        seed += 0x9E3779B97F4A7C15ULL;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    double nextDouble() { return static_cast<double>(nextLong() >> 11) * 0x1.0p-53; }
};

// Java (int) double: NaN -> 0, saturating, else truncate toward zero.
inline int32_t java_d2i(double x) {
    if (x != x) return 0;
    if (x >= 2147483647.0) return INT32_MAX;
    if (x <= -2147483648.0) return INT32_MIN;
    return static_cast<int32_t>(x);
}

// String.hashCode over UTF-16 code units (ASCII here): h = 31*h + c, int32 wrap.
inline int32_t java_string_hash(const char* s, size_t n) {
    uint32_t h = 0;
    for (size_t i = 0; i < n; i++) h = 31u * h + static_cast<unsigned char>(s[i]);
    return static_cast<int32_t>(h);
}
inline int32_t java_decimal_hash(int64_t ordinal) {
    std::string s = std::to_string(ordinal);  // Long.toString
    return java_string_hash(s.data(), s.size());
}

// -----------------------------------------------------------------------------
// java.util.HashMap<K,V> literal model (keys = int32 handles with an externally
// supplied String.hashCode; values = int64).  Reproduces: lazy table allocation,
// tableSizeFor, spread(), tail-append chains, value update in place, resize()
// lo/hi split preserving relative order, treeifyBin() -> resize() while the table
// is shorter than MIN_TREEIFY_CAPACITY (64), and — transliterated method by method
// from java.util.HashMap.TreeNode (JDK 21) — red-black TREE BINS: treeify,
// putTreeVal, find, balanceInsertion, rotateLeft/Right, moveRootToFront, split,
// untreeify.  HashIterator walks every bin through `next`, tree bins included, so
// the iteration order of a tree bin is its `next` list: treeifyBin keeps the chain
// order, treeify moves the root to the front, putTreeVal links a new node right
// behind its tree PARENT, split relinks in order.
// Keys are Strings in the reference: a tree bin orders by hash, then — String is
// Comparable — by compareTo; tieBreakOrder (identityHashCode) is unreachable for
// distinct Strings.  compareTo is only needed between different keys with EQUAL
// hashCode; it is computable when the ids are the decimal ordinals
// (`decimalKeys`), otherwise `unmodelled` is raised (the order is then unpinned).
// `treeified` records that some bin became a tree (what the product's kernels
// detect and hand to their host replay).  remove() is never used by the path.
// -----------------------------------------------------------------------------
struct JHashMap {
    struct Node {
        int32_t hash; int32_t key; int64_t val; int32_t next;
        // TreeNode fields (LinkedHashMap.Entry's before/after are unused by a plain HashMap)
        int32_t parent = -1, left = -1, right = -1, prev = -1;
        bool red = false, isTree = false;
    };
    static constexpr int32_t NIL = -1;
    static constexpr int TREEIFY_THRESHOLD = 8, UNTREEIFY_THRESHOLD = 6, MIN_TREEIFY_CAPACITY = 64;
    std::vector<Node> nodes;
    std::vector<int32_t> table;  // head node index per bucket, -1 = empty
    int32_t threshold = 0;
    int32_t size = 0;
    bool unmodelled = false;     // compareTo of two different keys with equal hashCode was needed and is unknown
    bool treeified = false;      // some bin was turned into a tree
    bool decimalKeys = false;    // keys are Long.toString(handle)

    static int32_t tableSizeFor(int32_t cap) {
        // n = -1 >>> numberOfLeadingZeros(cap - 1)
        uint32_t c = static_cast<uint32_t>(cap - 1);
        int nlz = (c == 0) ? 32 : __builtin_clz(c);
        int32_t n = static_cast<int32_t>(0xFFFFFFFFu >> (nlz & 31));  // Java shifts mod 32
        if (n < 0) return 1;
        if (n >= (1 << 30)) return 1 << 30;
        return n + 1;
    }
    static int32_t spread(int32_t h) {
        uint32_t u = static_cast<uint32_t>(h);
        return static_cast<int32_t>(u ^ (u >> 16));
    }

    explicit JHashMap(int32_t initialCapacity, bool decimal = false) : decimalKeys(decimal) {
        if (initialCapacity < 0) initialCapacity = 0;
        threshold = tableSizeFor(initialCapacity);  // HashMap(int): threshold holds initial cap
    }

    // compareComparables(String.class, k, x) = k.compareTo(x)
    int compareKeys(int32_t k, int32_t x) {
        if (!decimalKeys) { unmodelled = true; return 0; }
        const std::string a = std::to_string(k), b = std::to_string(x);
        const size_t lim = std::min(a.size(), b.size());
        for (size_t i = 0; i < lim; i++)
            if (a[i] != b[i]) return (int)(unsigned char)a[i] - (int)(unsigned char)b[i];
        return (int)a.size() - (int)b.size();
    }

    // ---- TreeNode methods ---------------------------------------------------------------------
    int32_t rootOf(int32_t r) const {
        for (int32_t p;;) { if ((p = nodes[r].parent) == NIL) return r; r = p; }
    }
    void moveRootToFront(int32_t root) {
        int32_t n;
        if (root != NIL && (n = (int32_t)table.size()) > 0) {
            int32_t index = (n - 1) & nodes[root].hash;
            int32_t first = table[index];
            if (root != first) {
                int32_t rn;
                table[index] = root;
                int32_t rp = nodes[root].prev;
                if ((rn = nodes[root].next) != NIL) nodes[rn].prev = rp;
                if (rp != NIL) nodes[rp].next = rn;
                if (first != NIL) nodes[first].prev = root;
                nodes[root].next = first;
                nodes[root].prev = NIL;
            }
        }
    }
    int32_t treeFind(int32_t start, int32_t h, int32_t k) {     // TreeNode.find(h, k, kc) with kc = String.class
        int32_t p = start;
        do {
            int32_t ph, pl = nodes[p].left, pr = nodes[p].right, q;
            int dir;
            if ((ph = nodes[p].hash) > h) p = pl;
            else if (ph < h) p = pr;
            else if (nodes[p].key == k) return p;
            else if (pl == NIL) p = pr;
            else if (pr == NIL) p = pl;
            else if ((dir = compareKeys(k, nodes[p].key)) != 0) p = (dir < 0) ? pl : pr;
            else if ((q = treeFind(pr, h, k)) != NIL) return q;
            else p = pl;
        } while (p != NIL);
        return NIL;
    }
    int32_t rotateLeft(int32_t root, int32_t p) {
        int32_t r, pp, rl;
        if (p != NIL && (r = nodes[p].right) != NIL) {
            if ((rl = nodes[p].right = nodes[r].left) != NIL) nodes[rl].parent = p;
            if ((pp = nodes[r].parent = nodes[p].parent) == NIL) nodes[root = r].red = false;
            else if (nodes[pp].left == p) nodes[pp].left = r;
            else nodes[pp].right = r;
            nodes[r].left = p;
            nodes[p].parent = r;
        }
        return root;
    }
    int32_t rotateRight(int32_t root, int32_t p) {
        int32_t l, pp, lr;
        if (p != NIL && (l = nodes[p].left) != NIL) {
            if ((lr = nodes[p].left = nodes[l].right) != NIL) nodes[lr].parent = p;
            if ((pp = nodes[l].parent = nodes[p].parent) == NIL) nodes[root = l].red = false;
            else if (nodes[pp].right == p) nodes[pp].right = l;
            else nodes[pp].left = l;
            nodes[l].right = p;
            nodes[p].parent = l;
        }
        return root;
    }
    int32_t balanceInsertion(int32_t root, int32_t x) {
        nodes[x].red = true;
        for (int32_t xp, xpp, xppl, xppr;;) {
            if ((xp = nodes[x].parent) == NIL) { nodes[x].red = false; return x; }
            else if (!nodes[xp].red || (xpp = nodes[xp].parent) == NIL) return root;
            if (xp == (xppl = nodes[xpp].left)) {
                if ((xppr = nodes[xpp].right) != NIL && nodes[xppr].red) {
                    nodes[xppr].red = false; nodes[xp].red = false; nodes[xpp].red = true;
                    x = xpp;
                } else {
                    if (x == nodes[xp].right) {
                        root = rotateLeft(root, x = xp);
                        xpp = (xp = nodes[x].parent) == NIL ? NIL : nodes[xp].parent;
                    }
                    if (xp != NIL) {
                        nodes[xp].red = false;
                        if (xpp != NIL) { nodes[xpp].red = true; root = rotateRight(root, xpp); }
                    }
                }
            } else {
                if (xppl != NIL && nodes[xppl].red) {
                    nodes[xppl].red = false; nodes[xp].red = false; nodes[xpp].red = true;
                    x = xpp;
                } else {
                    if (x == nodes[xp].left) {
                        root = rotateRight(root, x = xp);
                        xpp = (xp = nodes[x].parent) == NIL ? NIL : nodes[xp].parent;
                    }
                    if (xp != NIL) {
                        nodes[xp].red = false;
                        if (xpp != NIL) { nodes[xpp].red = true; root = rotateLeft(root, xpp); }
                    }
                }
            }
        }
    }
    void treeify(int32_t self) {                                // TreeNode.treeify(tab)
        int32_t root = NIL;
        for (int32_t x = self, next; x != NIL; x = next) {
            next = nodes[x].next;
            nodes[x].left = nodes[x].right = NIL;
            if (root == NIL) { nodes[x].parent = NIL; nodes[x].red = false; root = x; }
            else {
                int32_t k = nodes[x].key, h = nodes[x].hash;
                for (int32_t p = root;;) {
                    int dir; int32_t ph;
                    if ((ph = nodes[p].hash) > h) dir = -1;
                    else if (ph < h) dir = 1;
                    else if ((dir = compareKeys(k, nodes[p].key)) == 0) dir = -1;   // tieBreakOrder: not reproducible (unmodelled is set)
                    int32_t xp = p;
                    if ((p = (dir <= 0) ? nodes[p].left : nodes[p].right) == NIL) {
                        nodes[x].parent = xp;
                        if (dir <= 0) nodes[xp].left = x; else nodes[xp].right = x;
                        root = balanceInsertion(root, x);
                        break;
                    }
                }
            }
        }
        moveRootToFront(root);
    }
    void untreeify(int32_t self) {                              // replacementNode keeps the order; the nodes become plain
        for (int32_t q = self; q != NIL; q = nodes[q].next) {
            nodes[q].isTree = false; nodes[q].parent = nodes[q].left = nodes[q].right = nodes[q].prev = NIL; nodes[q].red = false;
        }
    }
    // returns the existing node for the key, or NIL after inserting a new TreeNode
    int32_t putTreeVal(int32_t self, int32_t h, int32_t k, int64_t v) {
        bool searched = false;
        int32_t root = (nodes[self].parent != NIL) ? rootOf(self) : self;
        for (int32_t p = root;;) {
            int dir; int32_t ph;
            if ((ph = nodes[p].hash) > h) dir = -1;
            else if (ph < h) dir = 1;
            else if (nodes[p].key == k) return p;
            else if ((dir = compareKeys(k, nodes[p].key)) == 0) {
                if (!searched) {
                    int32_t q, ch;
                    searched = true;
                    if (((ch = nodes[p].left) != NIL && (q = treeFind(ch, h, k)) != NIL) ||
                        ((ch = nodes[p].right) != NIL && (q = treeFind(ch, h, k)) != NIL))
                        return q;
                }
                dir = -1;                                       // tieBreakOrder: not reproducible (unmodelled is set)
            }
            int32_t xp = p;
            if ((p = (dir <= 0) ? nodes[p].left : nodes[p].right) == NIL) {
                int32_t xpn = nodes[xp].next;
                nodes.push_back(Node{h, k, v, xpn});            // newTreeNode(h, k, v, xpn)
                int32_t x = (int32_t)nodes.size() - 1;
                nodes[x].isTree = true;
                if (dir <= 0) nodes[xp].left = x; else nodes[xp].right = x;
                nodes[xp].next = x;
                nodes[x].parent = nodes[x].prev = xp;
                if (xpn != NIL) nodes[xpn].prev = x;
                moveRootToFront(balanceInsertion(root, x));
                return NIL;
            }
        }
    }
    void split(std::vector<int32_t>& tab, int32_t b, int32_t index, int32_t bit) {   // TreeNode.split
        int32_t loHead = NIL, loTail = NIL, hiHead = NIL, hiTail = NIL;
        int lc = 0, hc = 0;
        for (int32_t e = b, next; e != NIL; e = next) {
            next = nodes[e].next;
            nodes[e].next = NIL;
            if ((nodes[e].hash & bit) == 0) {
                if ((nodes[e].prev = loTail) == NIL) loHead = e; else nodes[loTail].next = e;
                loTail = e; ++lc;
            } else {
                if ((nodes[e].prev = hiTail) == NIL) hiHead = e; else nodes[hiTail].next = e;
                hiTail = e; ++hc;
            }
        }
        if (loHead != NIL) {
            if (lc <= UNTREEIFY_THRESHOLD) { untreeify(loHead); tab[index] = loHead; }
            else { tab[index] = loHead; if (hiHead != NIL) treeify(loHead); }   // (else is already treeified)
        }
        if (hiHead != NIL) {
            if (hc <= UNTREEIFY_THRESHOLD) { untreeify(hiHead); tab[index + bit] = hiHead; }
            else { tab[index + bit] = hiHead; if (loHead != NIL) treeify(hiHead); }
        }
    }
    void treeifyBin(int32_t hash) {
        int32_t n, index, e;
        if ((n = (int32_t)table.size()) < MIN_TREEIFY_CAPACITY) resize();
        else if ((e = table[index = (n - 1) & hash]) != NIL) {
            int32_t hd = NIL, tl = NIL;
            do {                                                // replacementTreeNode(e, null): same order, prev links added
                nodes[e].isTree = true; nodes[e].parent = nodes[e].left = nodes[e].right = NIL; nodes[e].red = false;
                nodes[e].prev = tl;
                if (tl == NIL) hd = e;
                tl = e;
            } while ((e = nodes[e].next) != NIL);
            table[index] = hd;
            treeified = true;
            treeify(hd);
        }
    }

    void resize() {
        int32_t oldCap = static_cast<int32_t>(table.size());
        int32_t oldThr = threshold;
        int32_t newCap, newThr = 0;
        if (oldCap > 0) {
            newCap = oldCap << 1;
            if (oldCap >= 16) newThr = oldThr << 1;
        } else if (oldThr > 0) {
            newCap = oldThr;
        } else {
            newCap = 16;
            newThr = 12;
        }
        if (newThr == 0) {
            float ft = static_cast<float>(newCap) * 0.75f;
            newThr = static_cast<int32_t>(ft);
        }
        threshold = newThr;
        std::vector<int32_t> newTab(static_cast<size_t>(newCap), -1);
        std::vector<int32_t> oldTab;
        oldTab.swap(table);
        table.swap(newTab);                                     // `table` IS the new table from here on (treeify -> moveRootToFront uses it)
        if (oldCap > 0) {
            for (int32_t j = 0; j < oldCap; j++) {
                int32_t e = oldTab[j];
                if (e < 0) continue;
                if (nodes[e].next == NIL) { table[nodes[e].hash & (newCap - 1)] = e; continue; }
                if (nodes[e].isTree) { split(table, e, j, oldCap); continue; }
                int32_t loHead = -1, loTail = -1, hiHead = -1, hiTail = -1;
                while (e >= 0) {
                    int32_t nx = nodes[e].next;
                    if ((nodes[e].hash & oldCap) == 0) {
                        if (loTail < 0) loHead = e; else nodes[loTail].next = e;
                        loTail = e;
                    } else {
                        if (hiTail < 0) hiHead = e; else nodes[hiTail].next = e;
                        hiTail = e;
                    }
                    e = nx;
                }
                if (loTail >= 0) { nodes[loTail].next = -1; table[j] = loHead; }
                if (hiTail >= 0) { nodes[hiTail].next = -1; table[j + oldCap] = hiHead; }
            }
        }
    }

    int32_t find(int32_t key, int32_t stringHash) {             // getNode
        if (table.empty()) return -1;
        int32_t h = spread(stringHash);
        int32_t first = table[(static_cast<int32_t>(table.size()) - 1) & h];
        if (first < 0) return -1;
        if (nodes[first].hash == h && nodes[first].key == key) return first;
        int32_t e = nodes[first].next;
        if (e < 0) return -1;
        if (nodes[first].isTree) return treeFind(nodes[first].parent != NIL ? rootOf(first) : first, h, key);   // getTreeNode
        do {
            if (nodes[e].hash == h && nodes[e].key == key) return e;
        } while ((e = nodes[e].next) >= 0);
        return -1;
    }

    // put(); returns true if a new mapping was created.
    bool put(int32_t key, int32_t stringHash, int64_t val) {
        int32_t h = spread(stringHash);
        if (table.empty()) resize();
        int32_t n = static_cast<int32_t>(table.size());
        int32_t i = (n - 1) & h;
        if (table[i] < 0) {
            nodes.push_back(Node{h, key, val, -1});
            table[i] = static_cast<int32_t>(nodes.size()) - 1;
        } else {
            int32_t p = table[i];
            int32_t e = -1;
            if (nodes[p].hash == h && nodes[p].key == key) e = p;
            else if (nodes[p].isTree) e = putTreeVal(p, h, key, val);
            else {
                for (int binCount = 0;; ++binCount) {
                    if ((e = nodes[p].next) < 0) {
                        nodes.push_back(Node{h, key, val, -1});
                        nodes[p].next = static_cast<int32_t>(nodes.size()) - 1;
                        if (binCount >= TREEIFY_THRESHOLD - 1) treeifyBin(h);  // -1 for 1st
                        break;
                    }
                    if (nodes[e].hash == h && nodes[e].key == key) break;
                    p = e;
                }
            }
            if (e >= 0) { nodes[e].val = val; return false; }   // existing mapping for key
        }
        if (++size > threshold) resize();
        return true;
    }

    template <class F> void forEach(F f) const {
        for (size_t b = 0; b < table.size(); b++)
            for (int32_t e = table[b]; e >= 0; e = nodes[e].next) f(nodes[e].key, nodes[e].val);
    }
};

// -----------------------------------------------------------------------------
// java.util.PriorityQueue<long[]>(comparingLong(a -> a[1])) literal model.
// -----------------------------------------------------------------------------
struct JPriorityQueue {
    struct E { int64_t idx; int64_t dist; };
    std::vector<E> q;
    static int cmp(const E& a, const E& b) { return (a.dist < b.dist) ? -1 : (a.dist > b.dist ? 1 : 0); }
    bool empty() const { return q.empty(); }
    void add(E x) {  // offer -> siftUp
        size_t k = q.size();
        q.push_back(x);
        while (k > 0) {
            size_t parent = (k - 1) >> 1;
            if (cmp(x, q[parent]) >= 0) break;
            q[k] = q[parent];
            k = parent;
        }
        q[k] = x;
    }
    E poll() {
        E result = q[0];
        size_t n = q.size() - 1;
        E x = q[n];
        q.pop_back();
        if (n > 0) {  // siftDown(0, x)
            size_t k = 0, half = n >> 1;
            while (k < half) {
                size_t child = (k << 1) + 1;
                size_t right = child + 1;
                if (right < n && cmp(q[child], q[right]) > 0) child = right;
                if (cmp(x, q[child]) <= 0) break;
                q[k] = q[child];
                k = child;
            }
            q[k] = x;
        }
        return result;
    }
};

// -----------------------------------------------------------------------------
// BitSet as uint64 words: bit i -> word i>>6, position i&63.
// -----------------------------------------------------------------------------
using Code = std::vector<uint64_t>;

inline int bitset_length(const uint64_t* w, int W) {
    for (int i = W - 1; i >= 0; i--)
        if (w[i]) return i * 64 + (64 - __builtin_clzll(w[i]));
    return 0;
}
inline bool bitset_get(const uint64_t* w, int W, int i) {
    int wi = i >> 6;
    return wi < W && ((w[wi] >> (i & 63)) & 1ULL);
}
// GreedyPartitioner.hamming (idx/GreedyPartitioner.java:78-82)
inline int64_t hamming(const uint64_t* a, const uint64_t* b, int W) {
    int64_t c = 0;
    for (int i = 0; i < W; i++) c += __builtin_popcountll(a[i] ^ b[i]);
    return c;
}
// GreedyPartitioner.computeKey (idx/GreedyPartitioner.java:87-96)
inline int64_t computeKey(const uint64_t* w, int W) {
    int64_t v = 0;
    int len = std::min(63, bitset_length(w, W));
    for (int i = 0; i < len; i++)
        if (bitset_get(w, W, i)) v |= (1LL << (62 - i));
    return v;
}

// -----------------------------------------------------------------------------
// Coding (idx/Coding.java)
// -----------------------------------------------------------------------------
// :342-347
inline double nextGaussian(SplittableRandom& r) {
    double u1 = std::max(4.9e-324 /* Double.MIN_VALUE */, r.nextDouble());
    double u2 = r.nextDouble();
    double mag = std::sqrt(-2.0 * std::log(u1));
    return mag * std::cos(2.0 * M_PI * u2);
}
// :349-353  (sequential, no FMA)
inline double dot(const double* a, const double* b, int d) {
    double acc = 0.0;
    for (int i = 0; i < d; i++) acc += a[i] * b[i];
    return acc;
}
// :355-361
inline bool vectorFinite(const double* v, int d) {
    for (int i = 0; i < d; i++)
        if (std::isnan(v[i]) || std::isinf(v[i])) return false;
    return true;
}

// alpha rows generation shared by buildRandomG (:141-151) and buildFromSample (:192-202)
void gen_alpha(SplittableRandom& rnd, int m, int d, double* alpha) {
    for (int j = 0; j < m; j++) {
        double norm = 0.0;
        for (int i = 0; i < d; i++) {
            double v = nextGaussian(rnd);
            alpha[(size_t)j * d + i] = v;
            norm += v * v;
        }
        norm = std::sqrt(std::max(1e-12, norm));
        for (int i = 0; i < d; i++) alpha[(size_t)j * d + i] /= norm;
    }
}

// Coding.buildRandomG (:136-161)
void buildRandomG(int d, int m, double omega, int64_t seed, double* alpha, double* r, double* w) {
    SplittableRandom rnd(seed);
    gen_alpha(rnd, m, d, alpha);
    for (int j = 0; j < m; j++) {
        r[j] = rnd.nextDouble() * omega;
        w[j] = omega;
    }
}

// Coding.buildFromSample (:184-241)
void buildFromSample(const double* sample, int ns, int d, int m, int64_t seed, double* alpha, double* r,
                     double* w) {
    SplittableRandom rnd(seed);
    gen_alpha(rnd, m, d, alpha);
    std::vector<double> mn(m, std::numeric_limits<double>::infinity());
    std::vector<double> mx(m, -std::numeric_limits<double>::infinity());
    for (int s = 0; s < ns; s++) {
        const double* v = sample + (size_t)s * d;
        for (int j = 0; j < m; j++) {
            double y = dot(v, alpha + (size_t)j * d, d);
            if (y < mn[j]) mn[j] = y;
            if (y > mx[j]) mx[j] = y;
        }
    }
    const double OMEGA_DIVISOR = 2.5;
    for (int j = 0; j < m; j++) {
        double range = std::max(1e-6, mx[j] - mn[j]);
        double omega = range / OMEGA_DIVISOR;
        if (!(omega > 0)) omega = 1e-3;
        w[j] = omega;
        r[j] = rnd.nextDouble() * omega;
    }
}

// Coding.H (:250-258)
void codingH(const double* v, int d, int m, const double* alpha, const double* r, const double* omega,
             int32_t* out) {
    for (int j = 0; j < m; j++) {
        double y = dot(v, alpha + (size_t)j * d, d) + r[j];
        out[j] = java_d2i(std::floor(y / omega[j]));
    }
}
// Coding.C (:285-301)
void codingC(const int32_t* H, int m, int lambda, uint64_t* words, int W) {
    for (int i = 0; i < W; i++) words[i] = 0;
    int pos = 0;
    for (int i = lambda - 1; i >= 0; i--) {
        for (int j = 0; j < m; j++) {
            uint32_t hj = static_cast<uint32_t>(H[j]) ^ 0x80000000u;
            if (((hj >> (i & 31)) & 1u) != 0) words[pos >> 6] |= (1ULL << (pos & 63));
            pos++;
        }
    }
}

// -----------------------------------------------------------------------------
// Oracle context
// -----------------------------------------------------------------------------
struct Partition {
    int64_t minKey, maxKey, centerKey;
    Code rep;
    std::vector<int32_t> ids;
};

struct Ctx {
    // paper.* / runtime.* knobs (cfg/SystemConfig.java:237-338)
    int T = 0, D = 0, m = 0, lambda = 0, d = 0;
    int blockSize = 64;       // PIS:92
    int defaultProbes = 5;    // PIS:93
    int cfgProbeOverride = -1;
    int maxGlobalCandidates = 20000;
    int refinementLimit = 20000;
    int hammingThreshold = 0;
    int TD() const { return T * D; }
    int bits() const { return m * lambda; }
    int W() const { return (bits() + 63) / 64; }

    std::vector<double> alpha, r, omega;            // [TD][m][d], [TD][m], [TD][m]
    std::vector<std::vector<Partition>> tables;     // [TD]
    std::vector<int32_t> javaHash;                  // per handle
    std::vector<uint8_t> deleted;                   // per handle
    std::vector<double> store;                      // plaintext store [n][d] (decrypt stand-in)
    std::vector<uint8_t> storeValid;                // 0 => loadPointIfActive()==null / decrypt error
    int64_t nIds = 0;
    bool frozen = false;
    bool decimalIds = false;              // ids are Long.toString(handle): String.compareTo between two ids is computable
    std::atomic<bool> unmodelled{false};  // a tree bin needed the order of two different ids with EQUAL hashCode and the ids' Strings
                                          // are not known here (non-decimal ids): iteration order not pinned
    std::atomic<bool> treeified{false};   // some HashMap turned a bin into a tree (modelled; what the product's kernels detect)

    // per-thread "last" fields are returned via out-params instead
};

int effectiveMaxProbes(const Ctx& c, int threadOverride) {  // PIS:880-888
    if (threadOverride > 0) return threadOverride;
    if (c.cfgProbeOverride > 0) return c.cfgProbeOverride;
    return c.defaultProbes;
}

// GreedyPartitioner.findNearestPartition (:101-124)
int findNearestPartition(const std::vector<Partition>& parts, int64_t qKey) {
    if (parts.empty()) return 0;
    int lo = 0, hi = (int)parts.size() - 1;
    while (lo <= hi) {
        int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        const Partition& p = parts[mid];
        if (qKey < p.minKey) hi = mid - 1;
        else if (qKey > p.maxKey) lo = mid + 1;
        else return mid;
    }
    if (lo <= 0) return 0;
    if (lo >= (int)parts.size()) return (int)parts.size() - 1;
    auto dist = [&](const Partition& p) -> int64_t {  // distanceToRange :126-130
        if (qKey < p.minKey) return p.minKey - qKey;
        if (qKey > p.maxKey) return qKey - p.maxKey;
        return 0;
    };
    int64_t dl = dist(parts[lo - 1]);
    int64_t dr = dist(parts[lo]);
    return (dl <= dr) ? (lo - 1) : lo;
}

// GreedyPartitioner.build (:37-76).  `order` = staged handles in insertion order,
// `codes` = [n][W] code of each staged element for this (t,d).
void greedyBuild(Ctx& c, const std::vector<int32_t>& order, const uint64_t* codes, int W,
                 std::vector<Partition>& out) {
    out.clear();
    if (order.empty()) return;
    // PIS:413-420: new HashMap<>(S.staged.size()); put(id, code) in staged order
    JHashMap idToCode((int32_t)order.size(), c.decimalIds);
    for (size_t i = 0; i < order.size(); i++) idToCode.put(order[i], c.javaHash[order[i]], (int64_t)i);
    if (idToCode.unmodelled) c.unmodelled = true;
    if (idToCode.treeified) c.treeified = true;
    struct Ent { int32_t id; int64_t key; int64_t src; };
    std::vector<Ent> ordered;
    ordered.reserve(order.size());
    idToCode.forEach([&](int32_t id, int64_t src) {
        ordered.push_back({id, computeKey(codes + (size_t)src * W, W), src});
    });
    // List.sort(comparingLong(value)) -> stable
    std::stable_sort(ordered.begin(), ordered.end(), [](const Ent& a, const Ent& b) { return a.key < b.key; });
    int bs = c.blockSize;
    for (size_t i = 0; i < ordered.size(); i += bs) {
        size_t end = std::min(i + (size_t)bs, ordered.size());
        Partition p;
        p.minKey = ordered[i].key;
        p.maxKey = ordered[end - 1].key;
        size_t mid = i + ((end - i - 1) >> 1);
        p.centerKey = ordered[mid].key;
        for (size_t j = i; j < end; j++) p.ids.push_back(ordered[j].id);
        const uint64_t* rc = codes + (size_t)ordered[mid].src * W;
        p.rep.assign(rc, rc + W);
        out.push_back(std::move(p));
    }
}

struct Cand { int32_t id; int64_t score; };

// PIS.collectPartitionOrdered (:726-753).  metadata.isDeleted(id) == deleted[id].
int collectPartitionOrdered(const Ctx& c, const Partition& p, const uint64_t* qBits, int W, JHashMap& best) {
    int newlySeen = 0;
    int64_t partDist = hamming(qBits, p.rep.data(), W);
    for (int32_t id : p.ids) {
        if (!c.deleted.empty() && c.deleted[id]) continue;
        int32_t e = best.find(id, c.javaHash[id]);
        if (e < 0 || partDist < best.nodes[e].val) {
            best.put(id, c.javaHash[id], partDist);
            newlySeen++;
        }
    }
    return newlySeen;
}

// Shared traversal of PIS.lookupCandidateIds (:459-582) and
// PIS.lookupCandidatesWithScores (:592-715).  qCodes = [TD][W].
// Returns entries stable-sorted by score in HashMap iteration order.
void routeTraverse(Ctx& c, const uint64_t* qCodes, int probes, std::vector<Cand>& out, int& rawSeen, bool* treeified = nullptr) {
    const int W = c.W();
    const int HARD_CAP = std::max(c.maxGlobalCandidates, c.refinementLimit);  // PIS:612-615
    JHashMap best(std::min(HARD_CAP, 1 << 16), c.decimalIds);                  // PIS:619
    rawSeen = 0;
    for (int t = 0; t < c.T && best.size < HARD_CAP; t++) {          // PIS:624
        for (int dv = 0; dv < c.D && best.size < HARD_CAP; dv++) {   // PIS:628
            const std::vector<Partition>& parts = c.tables[(size_t)t * c.D + dv];
            if (parts.empty()) continue;                              // PIS:638
            const uint64_t* qBits = qCodes + (size_t)(t * c.D + dv) * W;
            int64_t qKey = computeKey(qBits, W);                      // PIS:640
            int center = findNearestPartition(parts, qKey);           // PIS:641
            JPriorityQueue pq;                                        // PIS:643-644
            std::vector<uint8_t> visited(parts.size(), 0);
            pq.add({center, hamming(qBits, parts[center].rep.data(), W)});
            visited[center] = 1;
            int probesUsed = 0;
            while (!pq.empty() && probesUsed < probes && best.size < HARD_CAP) {  // PIS:657-659
                JPriorityQueue::E cur = pq.poll();
                int idx = (int)cur.idx;
                probesUsed++;
                rawSeen += collectPartitionOrdered(c, parts[idx], qBits, W, best);
                int left = idx - 1;
                if (left >= 0 && !visited[left]) {
                    visited[left] = 1;
                    pq.add({left, hamming(qBits, parts[left].rep.data(), W)});
                }
                int right = idx + 1;
                if (right < (int)parts.size() && !visited[right]) {
                    visited[right] = 1;
                    pq.add({right, hamming(qBits, parts[right].rep.data(), W)});
                }
            }
        }
    }
    if (best.unmodelled) c.unmodelled = true;
    if (best.treeified) c.treeified = true;
    if (treeified) *treeified = best.treeified;
    out.clear();
    out.reserve(best.size);
    best.forEach([&](int32_t id, int64_t v) { out.push_back({id, v}); });     // PIS:690-693
    std::stable_sort(out.begin(), out.end(), [](const Cand& a, const Cand& b) { return a.score < b.score; });
}

// QSI.l2 (:364-372)
inline double l2(const double* a, const double* b, int len) {
    double s = 0.0;
    for (int i = 0; i < len; i++) {
        double dd = a[i] - b[i];
        s += dd * dd;
    }
    return std::sqrt(s);
}
// QSI.isValid (:407-413)
inline bool isValid(const double* v, int d) {
    for (int i = 0; i < d; i++)
        if (!std::isfinite(v[i])) return false;
    return true;
}

struct SearchOut {
    std::vector<int32_t> ids;
    std::vector<double> dist;
    std::vector<int32_t> selected;  // F_q of the LAST pass (stage A.5 output, <= B)
    int candTotal = 0, candKept = 0, candDecrypted = 0, returned = 0, retried = 0;
    int lastReturned = 0;  // QSI.lastReturned (:318) — survives an empty retry pass
    std::vector<int32_t> touched;   // union over passes, in first-touch order
};

// QSI.search (:101-352) with the host decrypt replaced by the plaintext store.
void search(Ctx& c, const double* q, const uint64_t* qCodes, int K, int probeOverrideIn, int refineOverride,
            SearchOut& o) {
    o = SearchOut();
    if (!isValid(q, c.d)) return;  // :137-140
    bool retried = false;
    int probeOverride = probeOverrideIn;  // FSA:640-643 may have set a thread-local override
    while (true) {
        std::vector<Cand> cands;
        int rawSeen = 0;
        routeTraverse(c, qCodes, effectiveMaxProbes(c, probeOverride), cands, rawSeen);  // :153-154
        o.candTotal = rawSeen;                // :156
        o.candKept = (int)cands.size();       // :157
        if (cands.empty()) { o.ids.clear(); o.dist.clear(); o.returned = 0; o.selected.clear(); return; }  // :159
        // :161-165 re-sort is a stable no-op
        const int tau = c.hammingThreshold;
        const int runtimeLimit = (refineOverride > 0) ? refineOverride : c.refinementLimit;  // :170-171,460-463
        std::vector<int32_t> candidateIds;
        if (tau > 0) {  // :177-197
            for (const Cand& cd : cands)
                if (cd.score <= tau) {
                    candidateIds.push_back(cd.id);
                    if ((int)candidateIds.size() >= runtimeLimit) break;
                }
            if ((int)candidateIds.size() < runtimeLimit)
                for (const Cand& cd : cands)
                    if (cd.score > tau) {
                        candidateIds.push_back(cd.id);
                        if ((int)candidateIds.size() >= runtimeLimit) break;
                    }
        } else {  // :208-214
            for (const Cand& cd : cands) {
                candidateIds.push_back(cd.id);
                if ((int)candidateIds.size() >= runtimeLimit) break;
            }
        }
        o.selected = candidateIds;
        const int refineLimit = std::min((int)candidateIds.size(), runtimeLimit);  // :219
        struct Scored { int32_t id; double dist; };
        std::vector<Scored> scored;
        scored.reserve(refineLimit);
        for (int i = 0; i < refineLimit; i++) {  // :238-271
            int32_t id = candidateIds[i];
            // loadPointIfActive (PIS:717-724): deleted or missing -> null -> skipped
            if (!c.deleted.empty() && c.deleted[id]) continue;
            if (c.storeValid.empty() || !c.storeValid[id]) continue;
            const double* v = c.store.data() + (size_t)id * c.d;
            if (!isValid(v, c.d)) continue;  // :253-260
            scored.push_back({id, l2(q, v, c.d)});
            o.touched.push_back(id);  // touchedThisSession.add(id) (:263); deduplicated on return
        }
        o.candDecrypted = (int)scored.size();  // :274
        if (scored.empty()) { o.ids.clear(); o.dist.clear(); o.returned = 0; return; }  // :293
        // :298 Comparator.comparingDouble -> Double.compare, stable
        std::stable_sort(scored.begin(), scored.end(), [](const Scored& a, const Scored& b) {
            return a.dist < b.dist;  // no NaN after isValid; -0.0 cannot occur (sqrt of a sum of squares)
        });
        int eff = std::min(K, (int)scored.size());  // :300-307
        o.ids.clear();
        o.dist.clear();
        for (int i = 0; i < eff; i++) { o.ids.push_back(scored[i].id); o.dist.push_back(scored[i].dist); }
        o.returned = eff;
        o.lastReturned = eff;
        bool needRetry = (o.lastReturned < K) || (o.candDecrypted < 10 * K);  // :444-447
        if (!retried && needRetry) {  // :327-337
            retried = true;
            o.retried = 1;
            probeOverride = 10;
            continue;
        }
        std::sort(o.touched.begin(), o.touched.end());  // Set semantics
        o.touched.erase(std::unique(o.touched.begin(), o.touched.end()), o.touched.end());
        return;
    }
}

}  // namespace

// =============================================================================
// C API (ctypes)
// =============================================================================
extern "C" {

// ---- Java-semantics micro KAT hooks -----------------------------------------
uint64_t orc_splitmix_next(uint64_t* state) {
    SplittableRandom r((int64_t)*state);
    uint64_t v = r.nextLong();
    *state = r.seed;
    return v;
}
double orc_splitmix_next_double(uint64_t* state) {
    SplittableRandom r((int64_t)*state);
    double v = r.nextDouble();
    *state = r.seed;
    return v;
}
int32_t orc_d2i(double x) { return java_d2i(x); }
int32_t orc_string_hash(const char* s) { return java_string_hash(s, std::strlen(s)); }
int32_t orc_decimal_hash(int64_t ordinal) { return java_decimal_hash(ordinal); }
void orc_decimal_hashes(int64_t n, int32_t* out) {
    for (int64_t i = 0; i < n; i++) out[i] = java_decimal_hash(i);
}
int32_t orc_table_size_for(int32_t c) { return JHashMap::tableSizeFor(c); }

// HashMap iteration order of `n` keys inserted in the given order into new HashMap<>(initialCapacity).
// Returns flags: bit 0 = a bin was treeified at some point (modelled), bit 1 = unmodelled (String.compareTo between two
// different keys with equal hashCode was needed and the keys are not decimal ordinals).  decimalKeys: keys are Long.toString(key).
int orc_hashmap_order(int32_t initialCapacity, int64_t n, const int32_t* keys, const int32_t* hashes,
                      int32_t* out_keys, int32_t* out_final_cap, int decimalKeys) {
    JHashMap mp(initialCapacity, decimalKeys != 0);
    for (int64_t i = 0; i < n; i++) mp.put(keys[i], hashes[i], 0);
    int64_t k = 0;
    mp.forEach([&](int32_t key, int64_t) { out_keys[k++] = key; });
    if (out_final_cap) *out_final_cap = (int32_t)mp.table.size();
    return (mp.treeified ? 1 : 0) | (mp.unmodelled ? 2 : 0);
}

// PriorityQueue trace: ops[i] >= 0 => add(idx=i, dist=ops[i]); ops[i] == -1 => poll (writes idx to out).
int64_t orc_pq_trace(int64_t n, const int64_t* ops, int64_t* out) {
    JPriorityQueue pq;
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++) {
        if (ops[i] >= 0) pq.add({i, ops[i]});
        else if (!pq.empty()) out[k++] = pq.poll().idx;
    }
    return k;
}

// ---- GFunction generation ---------------------------------------------------
void orc_build_random_g(int d, int m, double omega, int64_t seed, double* alpha, double* r, double* w) {
    buildRandomG(d, m, omega, seed, alpha, r, w);
}
void orc_build_from_sample(const double* sample, int ns, int d, int m, int64_t seed, double* alpha, double* r,
                           double* w) {
    buildFromSample(sample, ns, d, m, seed, alpha, r, w);
}
// GFunctionRegistry.initialize (:110-126) + computeSeed (:291-293)
void orc_registry_init(const double* sample, int ns, int d, int m, int64_t baseSeed, int T, int D,
                       double* alpha, double* r, double* w) {
#pragma omp parallel for schedule(dynamic)
    for (int td = 0; td < T * D; td++) {
        int t = td / D, dv = td % D;
        int64_t seed = baseSeed + (int64_t)t * 1000003LL + dv;
        buildFromSample(sample, ns, d, m, seed, alpha + (size_t)td * m * d, r + (size_t)td * m,
                        w + (size_t)td * m);
    }
}

// ---- Coding -----------------------------------------------------------------
// returns -1 if v contains NaN/Inf (requireVector -> IllegalArgumentException)
int orc_H(const double* v, int d, int m, const double* alpha, const double* r, const double* omega,
          int32_t* out) {
    if (!vectorFinite(v, d)) return -1;
    codingH(v, d, m, alpha, r, omega, out);
    return 0;
}
int orc_C(const double* v, int d, int m, int lambda, const double* alpha, const double* r,
          const double* omega, uint64_t* words) {
    if (!vectorFinite(v, d)) return -1;
    std::vector<int32_t> H(m);
    codingH(v, d, m, alpha, r, omega, H.data());
    codingC(H.data(), m, lambda, words, (m * lambda + 63) / 64);
    return 0;
}
int64_t orc_compute_key(const uint64_t* words, int W) { return computeKey(words, W); }
int64_t orc_hamming(const uint64_t* a, const uint64_t* b, int W) { return hamming(a, b, W); }

// index/src/test/java/com/fspann/index/CodingQuickCheck.java:10-37 — the one
// property the reference pins: bit 0 of C(v) == bit (lambda-1) of H[0].
int orc_quickcheck(int32_t* H0_out, int* bit0_out) {
    const int d = 128, m = 24, lambda = 2;
    std::vector<double> v(d), alpha((size_t)m * d), r(m), w(m);
    for (int i = 0; i < d; i++) v[i] = i * 0.01;
    buildRandomG(d, m, 1.0, 12345LL, alpha.data(), r.data(), w.data());
    std::vector<int32_t> H(m);
    codingH(v.data(), d, m, alpha.data(), r.data(), w.data(), H.data());
    uint64_t words[1];
    codingC(H.data(), m, lambda, words, 1);
    int expected = (int)((static_cast<uint32_t>(H[0]) >> (lambda - 1)) & 1u);
    int actual = (int)(words[0] & 1ULL);
    if (H0_out) *H0_out = H[0];
    if (bit0_out) *bit0_out = actual;
    return expected == actual ? 0 : 1;
}

// ---- Context ----------------------------------------------------------------
void* orc_ctx_create(int T, int D, int m, int lambda, int d, int maxGlobalCandidates, int refinementLimit,
                     int cfgProbeOverride, int hammingThreshold) {
    Ctx* c = new Ctx();
    c->T = T; c->D = D; c->m = m; c->lambda = lambda; c->d = d;
    c->maxGlobalCandidates = maxGlobalCandidates;
    c->refinementLimit = refinementLimit;
    c->cfgProbeOverride = cfgProbeOverride;
    c->hammingThreshold = hammingThreshold;
    c->tables.resize((size_t)T * D);
    return c;
}
void orc_ctx_destroy(void* p) { delete static_cast<Ctx*>(p); }
int orc_unmodelled(void* p) { return static_cast<Ctx*>(p)->unmodelled ? 1 : 0; }
int orc_treeified(void* p) { return static_cast<Ctx*>(p)->treeified ? 1 : 0; }

void orc_set_gfunctions(void* p, const double* alpha, const double* r, const double* omega) {
    Ctx* c = static_cast<Ctx*>(p);
    size_t TD = c->TD();
    c->alpha.assign(alpha, alpha + TD * c->m * c->d);
    c->r.assign(r, r + TD * c->m);
    c->omega.assign(omega, omega + TD * c->m);
}

// id metadata: javaHash per handle (NULL => decimal ordinals), deleted flags (NULL => none)
void orc_set_id_meta(void* p, int64_t n, const int32_t* javaHash, const uint8_t* deleted) {
    Ctx* c = static_cast<Ctx*>(p);
    c->nIds = n;
    c->javaHash.resize(n);
    c->decimalIds = (javaHash == nullptr);
    c->unmodelled = false;   // new hashCodes: what earlier maps did says nothing about the ones to come
    c->treeified = false;
    if (javaHash) std::copy(javaHash, javaHash + n, c->javaHash.begin());
    else for (int64_t i = 0; i < n; i++) c->javaHash[i] = java_decimal_hash(i);
    if (deleted) c->deleted.assign(deleted, deleted + n); else c->deleted.clear();
}

// plaintext store = stand-in for loadPointIfActive + decryptFromPoint
void orc_set_store(void* p, int64_t n, const double* vecs, const uint8_t* valid) {
    Ctx* c = static_cast<Ctx*>(p);
    c->store.assign(vecs, vecs + (size_t)n * c->d);
    if (valid) c->storeValid.assign(valid, valid + n); else c->storeValid.assign(n, 1);
}

// TokenGen math (QueryTokenFactory.create :98-131): codes[TD][W]; -1 on NaN/Inf
int orc_encode(void* p, int64_t nq, const double* q, uint64_t* codes) {
    Ctx* c = static_cast<Ctx*>(p);
    const int W = c->W(), TD = c->TD();
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t i = 0; i < nq; i++) {
        const double* v = q + (size_t)i * c->d;
        if (!vectorFinite(v, c->d)) { bad |= 1; continue; }
        std::vector<int32_t> H(c->m);
        for (int td = 0; td < TD; td++) {
            codingH(v, c->d, c->m, c->alpha.data() + (size_t)td * c->m * c->d, c->r.data() + (size_t)td * c->m,
                    c->omega.data() + (size_t)td * c->m, H.data());
            codingC(H.data(), c->m, c->lambda, codes + ((size_t)i * TD + td) * W, W);
        }
    }
    return bad ? -1 : 0;
}
int orc_hashes(void* p, int64_t nq, const double* q, int32_t* H) {  // [nq][TD][m]
    Ctx* c = static_cast<Ctx*>(p);
    for (int64_t i = 0; i < nq; i++)
        for (int td = 0; td < c->TD(); td++)
            codingH(q + (size_t)i * c->d, c->d, c->m, c->alpha.data() + (size_t)td * c->m * c->d,
                    c->r.data() + (size_t)td * c->m, c->omega.data() + (size_t)td * c->m,
                    H + ((size_t)i * c->TD() + td) * c->m);
    return 0;
}

// Setup: PIS.insert staging order + finalizeForSearch + build (PIS:265-347,372-434,789-845).
// `order[n]` = handles in the order they reach `staged` (for the stock pipeline:
// 999,1000,...,N-1,0,...,998, see orc_staged_order); `codes` = [n][TD][W] indexed by POSITION in order.
void orc_build_index(void* p, int64_t n, const int32_t* order, const uint64_t* codes) {
    Ctx* c = static_cast<Ctx*>(p);
    const int W = c->W(), TD = c->TD();
    std::vector<int32_t> ord(order, order + n);
    int unm = 0;
#pragma omp parallel for schedule(dynamic) reduction(| : unm)
    for (int td = 0; td < TD; td++) {
        std::vector<uint64_t> cw((size_t)n * W);
        for (int64_t i = 0; i < n; i++)
            for (int w = 0; w < W; w++) cw[(size_t)i * W + w] = codes[((size_t)i * TD + td) * W + w];
        Ctx tmp;  // private flag holder to stay race-free
        tmp.blockSize = c->blockSize;
        tmp.javaHash = c->javaHash;
        tmp.decimalIds = c->decimalIds;
        greedyBuild(tmp, ord, cw.data(), W, c->tables[td]);
        unm |= (tmp.unmodelled ? 1 : 0) | (tmp.treeified ? 2 : 0);
    }
    if (unm & 1) c->unmodelled = true;
    if (unm & 2) c->treeified = true;
    c->frozen = true;
}
// Staged order of the stock pipeline: first MIN_SAMPLE_SIZE-1 ids parked, flushed last.
void orc_staged_order(int64_t n, int64_t minSample, int32_t* out) {
    int64_t k = 0;
    if (n < minSample) { for (int64_t i = 0; i < n; i++) out[k++] = (int32_t)i; return; }
    for (int64_t i = minSample - 1; i < n; i++) out[k++] = (int32_t)i;
    for (int64_t i = 0; i < minSample - 1; i++) out[k++] = (int32_t)i;
}

// Import a frozen table (same SoA the product's fspann_set_index takes)
void orc_set_index(void* p, int td, int64_t nparts, const int64_t* minKey, const int64_t* maxKey,
                   const uint64_t* rep, const int64_t* idOff, const int32_t* ids) {
    Ctx* c = static_cast<Ctx*>(p);
    const int W = c->W();
    std::vector<Partition>& parts = c->tables[td];
    parts.clear();
    parts.resize(nparts);
    for (int64_t i = 0; i < nparts; i++) {
        parts[i].minKey = minKey[i];
        parts[i].maxKey = maxKey[i];
        parts[i].centerKey = 0;
        parts[i].rep.assign(rep + (size_t)i * W, rep + (size_t)(i + 1) * W);
        parts[i].ids.assign(ids + idOff[i], ids + idOff[i + 1]);
    }
    c->frozen = true;
}
int64_t orc_index_nparts(void* p, int td) { return (int64_t)static_cast<Ctx*>(p)->tables[td].size(); }
int64_t orc_index_nids(void* p, int td) {
    int64_t n = 0;
    for (auto& pt : static_cast<Ctx*>(p)->tables[td]) n += (int64_t)pt.ids.size();
    return n;
}
void orc_get_index(void* p, int td, int64_t* minKey, int64_t* maxKey, uint64_t* rep, int64_t* idOff,
                   int32_t* ids) {
    Ctx* c = static_cast<Ctx*>(p);
    const int W = c->W();
    int64_t off = 0;
    const auto& parts = c->tables[td];
    for (size_t i = 0; i < parts.size(); i++) {
        minKey[i] = parts[i].minKey;
        maxKey[i] = parts[i].maxKey;
        for (int w = 0; w < W; w++) rep[i * W + w] = parts[i].rep[w];
        idOff[i] = off;
        for (int32_t id : parts[i].ids) ids[off++] = id;
    }
    idOff[parts.size()] = off;
}

// Route: lookupCandidatesWithScores (truncate=0) / lookupCandidateIds (truncate=1, PIS:558-565).
// Outputs are [nq][cap] row-major; count[q] entries valid.  Returns max count (so callers can size cap).
int64_t orc_route(void* p, int64_t nq, const uint64_t* codes, int probeOverride, int truncate, int64_t cap,
                  int32_t* ids, int32_t* score, int32_t* count, int32_t* rawSeen) {
    Ctx* c = static_cast<Ctx*>(p);
    const int W = c->W(), TD = c->TD();
    const int HARD_CAP = std::max(c->maxGlobalCandidates, c->refinementLimit);
    int64_t mx = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(max : mx)
    for (int64_t i = 0; i < nq; i++) {
        std::vector<Cand> out;
        int raw = 0;
        Ctx& cc = *c;
        routeTraverse(cc, codes + (size_t)i * TD * W, effectiveMaxProbes(cc, probeOverride), out, raw);
        int64_t n = (int64_t)out.size();
        if (truncate && n > HARD_CAP) n = HARD_CAP;
        if (n > mx) mx = n;
        if (count) count[i] = (int32_t)n;
        if (rawSeen) rawSeen[i] = raw;
        for (int64_t k = 0; k < n && k < cap; k++) {
            if (ids) ids[(size_t)i * cap + k] = out[k].id;
            if (score) score[(size_t)i * cap + k] = (int32_t)out[k].score;
        }
    }
    return mx;
}

// Per query: did the literal HashMap model meet a treeifyBin() on a table >= 64 (iteration order unmodelled)?
void orc_route_treeified(void* p, int64_t nq, const uint64_t* codes, int probeOverride, uint8_t* flags) {
    Ctx* c = static_cast<Ctx*>(p);
    const int W = c->W(), TD = c->TD();
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t i = 0; i < nq; i++) {
        std::vector<Cand> out;
        int raw = 0;
        bool t = false;
        routeTraverse(*c, codes + (size_t)i * TD * W, effectiveMaxProbes(*c, probeOverride), out, raw, &t);
        flags[i] = t ? 1 : 0;
    }
}

// Full search for a batch.  out_ids/out_dist = [nq][K]; sel = [nq][selCap] (stage A.5 list of last pass).
// metrics = [nq][5] {candTotal, candKept, candDecrypted, returned, retried}.
void orc_search(void* p, int64_t nq, const double* q, const uint64_t* codes, int K, int probeOverride,
                int refineOverride, int32_t* out_ids, double* out_dist, int32_t* out_count, int32_t* sel,
                int32_t* sel_count, int64_t selCap, int32_t* metrics, int threads) {
    Ctx* c = static_cast<Ctx*>(p);
    const int W = c->W(), TD = c->TD();
#if defined(_OPENMP)
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t i = 0; i < nq; i++) {
        SearchOut o;
        search(*c, q + (size_t)i * c->d, codes + (size_t)i * TD * W, K, probeOverride, refineOverride, o);
        if (out_count) out_count[i] = o.returned;
        for (int k = 0; k < K; k++) {
            if (out_ids) out_ids[(size_t)i * K + k] = k < o.returned ? o.ids[k] : -1;
            if (out_dist) out_dist[(size_t)i * K + k] = k < o.returned ? o.dist[k] : std::numeric_limits<double>::infinity();
        }
        if (sel_count) sel_count[i] = (int32_t)o.selected.size();
        if (sel)
            for (int64_t k = 0; k < (int64_t)o.selected.size() && k < selCap; k++) sel[(size_t)i * selCap + k] = o.selected[k];
        if (metrics) {
            metrics[i * 5 + 0] = o.candTotal;
            metrics[i * 5 + 1] = o.candKept;
            metrics[i * 5 + 2] = o.candDecrypted;
            metrics[i * 5 + 3] = o.lastReturned;
            metrics[i * 5 + 4] = o.retried;
        }
    }
}

// Stage B (distance part) + C on packed candidates: cand = [nq][B][d] doubles, cand_count[q] valid rows.
void orc_refine(int64_t nq, int d, int64_t B, const double* q, const double* cand, const int32_t* cand_ids,
                const int32_t* cand_count, int K, int32_t* out_ids, double* out_dist, int32_t* out_count) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; i++) {
        struct Scored { int32_t id; double dist; };
        std::vector<Scored> scored;
        for (int j = 0; j < cand_count[i]; j++) {
            const double* v = cand + ((size_t)i * B + j) * d;
            if (!isValid(v, d)) continue;
            scored.push_back({cand_ids[(size_t)i * B + j], l2(q + (size_t)i * d, v, d)});
        }
        std::stable_sort(scored.begin(), scored.end(), [](const Scored& a, const Scored& b) { return a.dist < b.dist; });
        int eff = std::min(K, (int)scored.size());
        out_count[i] = eff;
        for (int k = 0; k < K; k++) {
            out_ids[(size_t)i * K + k] = k < eff ? scored[k].id : -1;
            out_dist[(size_t)i * K + k] = k < eff ? scored[k].dist : std::numeric_limits<double>::infinity();
        }
    }
}

// GroundtruthPrecompute (api/.../GroundtruthPrecompute.java:142-189,218-272): squared distance with a FLOAT subtraction per
// dimension (q[i] and v are floats: Java's binary numeric promotion), fp64 squares summed in order; the k smallest by
// (distance, id) — HeapK keeps exactly those, idsAscending() orders them by BY_D_THEN_ID.
void orc_groundtruth(int64_t n, const float* base, int64_t nq, const float* q, int d, int k, int32_t* out_ids, double* out_d2) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t qi = 0; qi < nq; qi++) {
        std::vector<std::pair<double, int32_t>> all((size_t)n);
        const float* qr = q + (size_t)qi * d;
        for (int64_t r = 0; r < n; r++) {
            const float* v = base + (size_t)r * d;
            double sum = 0.0;
            for (int i = 0; i < d; i++) {
                volatile float df = qr[i] - v[i];      // float arithmetic, rounded to float (no excess precision)
                double dd = (double)df;
                sum += dd * dd;
            }
            all[(size_t)r] = {sum, (int32_t)r};
        }
        const int64_t kk = std::min<int64_t>(k, n);
        std::partial_sort(all.begin(), all.begin() + kk, all.end());     // pair order = (distance, id): Comparator BY_D_THEN_ID
        for (int64_t j = 0; j < k; j++) {
            out_ids[(size_t)qi * k + j] = j < kk ? all[(size_t)j].second : -1;
            if (out_d2) out_d2[(size_t)qi * k + j] = j < kk ? all[(size_t)j].first : std::numeric_limits<double>::infinity();
        }
    }
}

// ForwardSecureANNSystem.computeMetricsAtK (FSA:770-835) with BaseVectorReader.l2 (FSA:1017-1073).
void orc_metrics(int64_t n, const float* base, int64_t nq, const float* q, int d, int k, const int32_t* ann, int64_t ann_stride,
                 const int32_t* ann_count, const int32_t* gt, int64_t gt_stride, double* recall, double* ratio) {
    auto l2 = [&](const float* qr, int32_t id) {
        double sum = 0.0;
        for (int i = 0; i < d; i++) {
            double dd = (double)qr[i] - (double)base[(size_t)id * d + i];
            sum += dd * dd;
        }
        return std::sqrt(sum);
    };
    for (int64_t qi = 0; qi < nq; qi++) {
        const int na = ann_count ? std::max(0, std::min<int>(ann_count[qi], (int)ann_stride)) : (int)ann_stride;
        const int32_t* a = ann + (size_t)qi * ann_stride;
        const int32_t* g = gt + (size_t)qi * gt_stride;
        int hits = 0;
        for (int i = 0; i < std::min(k, na); i++) {
            bool in = false;
            for (int j = 0; j < k; j++) in = in || g[j] == a[i];
            hits += in;
        }
        recall[qi] = hits / (double)k;
        double r = std::numeric_limits<double>::quiet_NaN();
        if (na >= k) {
            double sum = 0.0;
            int used = 0;
            for (int i = 0; i < k; i++) {
                if (a[i] < 0 || a[i] >= n || g[i] < 0 || g[i] >= n) continue;
                double dGt = l2(q + (size_t)qi * d, g[i]);
                if (dGt <= 0) continue;
                sum += l2(q + (size_t)qi * d, a[i]) / dGt;
                used++;
            }
            if (used == k) r = sum / k;
        }
        ratio[qi] = r;
    }
}

int orc_num_threads() {
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
