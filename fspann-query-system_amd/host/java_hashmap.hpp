// java_hashmap.hpp — literal host model of java.util.HashMap<String, Long> (JDK 21), INCLUDING red-black tree bins.
// Product code (pure host C++17, no HIP): used by the rare paths of libfspann_hip.so that must reproduce the JVM's
// iteration order when the closed form of the kernels (bin at the final table length, then first insertion — exact while
// every bin is a plain chain) does not apply:
//   * Route: a query whose HashMap<String,Long> bestScore (PIS:619,690-693) would treeify a bin is replayed put by put
//     (route_replay.hpp) instead of being refused;
//   * Setup: a staging map HashMap<String,BitSet>(staged.size()) (PIS:413, idx/GreedyPartitioner.java:45-48) that
//     treeifies a bin gets its iteration order from this model instead of the GPU's (bin, position) sort.
// Never includes or links anything under oracle/ (which holds its own, independently written model).
//
// What is modelled (java.util.HashMap): lazy table allocation, tableSizeFor, hash() spreading, tail-append chains, value
// update in place, resize() with order-preserving lo/hi split, treeifyBin() (resize while the table is shorter than
// MIN_TREEIFY_CAPACITY, else TreeNodes), TreeNode.treeify / putTreeVal / balanceInsertion / rotateLeft / rotateRight /
// moveRootToFront, TreeNode.split with untreeify at UNTREEIFY_THRESHOLD, and iteration through the `next` links (which is
// how HashIterator walks tree bins too).  Keys are Strings in the reference; here a key is an int32 handle with its
// String.hashCode, and `KeyOrder` answers String.compareTo for two handles whose hashCodes are EQUAL (the only time the
// JVM looks at it: tree bins order by hash first, comparableClassFor(String) then compareTo; tieBreakOrder is never
// reached for distinct Strings).  remove() is not modelled: the path never removes.
#pragma once
#include <cstdint>
#include <vector>

namespace fspann {
namespace jdk {

// String.compareTo(Long.toString(a), Long.toString(b)) for non-negative a, b: UTF-16 code units left to right, then length.
inline int compare_decimal_strings(int64_t a, int64_t b) {
    char sa[24], sb[24];
    int na = 0, nb = 0;
    auto put = [](int64_t v, char* s) { char t[24]; int n = 0; do { t[n++] = static_cast<char>('0' + v % 10); v /= 10; } while (v); for (int i = 0; i < n; i++) s[i] = t[n - 1 - i]; return n; };
    na = put(a, sa);
    nb = put(b, sb);
    const int lim = na < nb ? na : nb;
    for (int i = 0; i < lim; i++)
        if (sa[i] != sb[i]) return static_cast<int>(sa[i]) - static_cast<int>(sb[i]);
    return na - nb;
}

// KeyOrder: int operator()(int32_t a, int32_t b) const -> sign of a.compareTo(b); 0 = "unknown" (raises `unmodelled`).
template <class KeyOrder>
class HashMapModel {
  public:
    static constexpr int kTreeifyThreshold = 8, kUntreeifyThreshold = 6, kMinTreeifyCapacity = 64;
    struct Node {
        int32_t hash;      // spread hash
        int32_t key;
        int64_t val;
        int32_t next, prev;             // `next` is the iteration order inside a bin (chains and trees alike)
        int32_t parent, left, right;    // tree links (tree bins only)
        bool red, tree;
    };
    bool unmodelled = false;   // a tree bin had to order two different keys with equal hashCode and KeyOrder did not know
    bool treeified = false;    // at least one bin became a tree at some point (diagnostics)

    HashMapModel(int32_t initialCapacity, KeyOrder order) : order_(order) {
        if (initialCapacity < 0) initialCapacity = 0;
        threshold_ = table_size_for(initialCapacity);      // HashMap(int): the threshold field holds the initial capacity
    }
    static int32_t table_size_for(int32_t cap) {
        const uint32_t c = static_cast<uint32_t>(cap - 1);
        const int nlz = (c == 0) ? 32 : __builtin_clz(c);
        const int32_t n = static_cast<int32_t>(0xFFFFFFFFu >> (nlz & 31));
        if (n < 0) return 1;
        if (n >= (1 << 30)) return 1 << 30;
        return n + 1;
    }
    static int32_t spread(int32_t h) { const uint32_t u = static_cast<uint32_t>(h); return static_cast<int32_t>(u ^ (u >> 16)); }
    int32_t size() const { return size_; }
    int32_t capacity() const { return static_cast<int32_t>(tab_.size()); }
    void reserve(size_t n) { nodes_.reserve(n); }

    // Map.get: pointer to the value or nullptr
    int64_t* get(int32_t key, int32_t stringHash) {
        const int32_t e = find_node(key, spread(stringHash));
        return e < 0 ? nullptr : &nodes_[e].val;
    }
    // Map.put; returns true when a new mapping was created
    bool put(int32_t key, int32_t stringHash, int64_t val) {
        const int32_t h = spread(stringHash);
        if (tab_.empty()) resize();
        const int32_t n = static_cast<int32_t>(tab_.size());
        const int32_t i = (n - 1) & h;
        int32_t p = tab_[i];
        if (p < 0) {
            tab_[i] = new_node(h, key, val);
        } else {
            int32_t e = -1;
            if (nodes_[p].hash == h && nodes_[p].key == key) e = p;
            else if (nodes_[p].tree) e = put_tree_val(p, h, key, val);
            else {
                for (int binCount = 0;; ++binCount) {
                    e = nodes_[p].next;
                    if (e < 0) {
                        const int32_t x = new_node(h, key, val);
                        nodes_[p].next = x;
                        if (binCount >= kTreeifyThreshold - 1) treeify_bin(h);
                        break;
                    }
                    if (nodes_[e].hash == h && nodes_[e].key == key) break;
                    p = e;
                }
            }
            if (e >= 0) { nodes_[e].val = val; return false; }      // existing mapping: value replaced, position kept
        }
        if (++size_ > threshold_) resize();
        return true;
    }
    // HashIterator order: bins ascending, `next` links inside a bin
    template <class F> void for_each(F f) const {
        for (size_t b = 0; b < tab_.size(); b++)
            for (int32_t e = tab_[b]; e >= 0; e = nodes_[e].next) f(nodes_[e].key, nodes_[e].val);
    }

  private:
    std::vector<Node> nodes_;
    std::vector<int32_t> tab_;
    int32_t threshold_ = 0, size_ = 0;
    KeyOrder order_;

    int32_t new_node(int32_t h, int32_t key, int64_t val) {
        nodes_.push_back(Node{h, key, val, -1, -1, -1, -1, -1, false, false});
        return static_cast<int32_t>(nodes_.size()) - 1;
    }
    // dir of (h, key) relative to node p, as treeify / putTreeVal / find compute it: hash first, then String.compareTo
    int dir_of(int32_t h, int32_t key, int32_t p) {
        const int32_t ph = nodes_[p].hash;
        if (ph > h) return -1;
        if (ph < h) return 1;
        const int c = order_(key, nodes_[p].key);
        if (c == 0) { unmodelled = true; return -1; }      // tieBreakOrder (identityHashCode) is not reproducible outside the JVM
        return c < 0 ? -1 : 1;
    }
    int32_t find_node(int32_t key, int32_t h) const {
        if (tab_.empty()) return -1;
        int32_t e = tab_[(static_cast<int32_t>(tab_.size()) - 1) & h];
        if (e < 0) return -1;
        if (!nodes_[e].tree) {
            for (; e >= 0; e = nodes_[e].next)
                if (nodes_[e].hash == h && nodes_[e].key == key) return e;
            return -1;
        }
        // TreeNode.getTreeNode: the search's RESULT does not depend on the path taken — walk the bin's `next` list
        for (; e >= 0; e = nodes_[e].next)
            if (nodes_[e].hash == h && nodes_[e].key == key) return e;
        return -1;
    }
    void resize() {
        const int32_t oldCap = static_cast<int32_t>(tab_.size());
        const int32_t oldThr = threshold_;
        int32_t newCap, newThr = 0;
        if (oldCap > 0) {
            if (oldCap >= (1 << 30)) { threshold_ = INT32_MAX; return; }
            newCap = oldCap << 1;
            if (newCap < (1 << 30) && oldCap >= 16) newThr = oldThr << 1;
        } else if (oldThr > 0) newCap = oldThr;
        else { newCap = 16; newThr = 12; }
        if (newThr == 0) {
            const float ft = static_cast<float>(newCap) * 0.75f;
            newThr = (newCap < (1 << 30) && ft < static_cast<float>(1 << 30)) ? static_cast<int32_t>(ft) : INT32_MAX;
        }
        threshold_ = newThr;
        std::vector<int32_t> old(static_cast<size_t>(newCap), -1);
        old.swap(tab_);                                     // tab_ = new table, old = old table
        for (int32_t j = 0; j < oldCap; j++) {
            int32_t e = old[j];
            if (e < 0) continue;
            if (nodes_[e].next < 0) { tab_[nodes_[e].hash & (newCap - 1)] = e; continue; }
            if (nodes_[e].tree) { split(e, j, oldCap); continue; }
            int32_t loHead = -1, loTail = -1, hiHead = -1, hiTail = -1;
            while (e >= 0) {
                const int32_t nx = nodes_[e].next;
                if ((nodes_[e].hash & oldCap) == 0) { if (loTail < 0) loHead = e; else nodes_[loTail].next = e; loTail = e; }
                else { if (hiTail < 0) hiHead = e; else nodes_[hiTail].next = e; hiTail = e; }
                e = nx;
            }
            if (loTail >= 0) { nodes_[loTail].next = -1; tab_[j] = loHead; }
            if (hiTail >= 0) { nodes_[hiTail].next = -1; tab_[j + oldCap] = hiHead; }
        }
    }
    // HashMap.treeifyBin
    void treeify_bin(int32_t hash) {
        const int32_t n = static_cast<int32_t>(tab_.size());
        if (n < kMinTreeifyCapacity) { resize(); return; }
        const int32_t index = (n - 1) & hash;
        int32_t e = tab_[index];
        if (e < 0) return;
        int32_t tl = -1;
        for (; e >= 0; e = nodes_[e].next) {                // replacementTreeNode keeps the order of the chain
            Node& p = nodes_[e];
            p.tree = true; p.prev = tl; p.parent = p.left = p.right = -1; p.red = false;
            tl = e;
        }
        treeified = true;
        treeify(tab_[index]);
    }
    // TreeNode.treeify: build the tree in `next` order, then move the root to the front of the bin
    void treeify(int32_t head) {
        int32_t root = -1;
        for (int32_t x = head, next; x >= 0; x = next) {
            next = nodes_[x].next;
            nodes_[x].left = nodes_[x].right = -1;
            if (root < 0) { nodes_[x].parent = -1; nodes_[x].red = false; root = x; continue; }
            const int32_t h = nodes_[x].hash, k = nodes_[x].key;
            for (int32_t p = root;;) {
                const int dir = dir_of(h, k, p);
                const int32_t xp = p;
                p = (dir <= 0) ? nodes_[p].left : nodes_[p].right;
                if (p < 0) {
                    nodes_[x].parent = xp;
                    if (dir <= 0) nodes_[xp].left = x; else nodes_[xp].right = x;
                    root = balance_insertion(root, x);
                    break;
                }
            }
        }
        move_root_to_front(root);
    }
    // TreeNode.putTreeVal: returns the existing node, or -1 after linking a new one
    int32_t put_tree_val(int32_t first, int32_t h, int32_t k, int64_t v) {
        int32_t root = first;
        while (nodes_[root].parent >= 0) root = nodes_[root].parent;
        for (int32_t p = root;;) {
            if (nodes_[p].hash == h && nodes_[p].key == k) return p;
            const int dir = dir_of(h, k, p);
            const int32_t xp = p;
            p = (dir <= 0) ? nodes_[p].left : nodes_[p].right;
            if (p < 0) {
                const int32_t xpn = nodes_[xp].next;
                const int32_t x = new_node(h, k, v);        // may reallocate nodes_: no references held across it
                nodes_[x].tree = true;
                nodes_[x].next = xpn;
                if (dir <= 0) nodes_[xp].left = x; else nodes_[xp].right = x;
                nodes_[xp].next = x;                        // the new node follows its tree parent in iteration order
                nodes_[x].parent = nodes_[x].prev = xp;
                if (xpn >= 0) nodes_[xpn].prev = x;
                move_root_to_front(balance_insertion(root, x));
                return -1;
            }
        }
    }
    void move_root_to_front(int32_t root) {
        if (root < 0 || tab_.empty()) return;
        const int32_t index = (static_cast<int32_t>(tab_.size()) - 1) & nodes_[root].hash;
        const int32_t first = tab_[index];
        if (root == first) return;
        tab_[index] = root;
        const int32_t rp = nodes_[root].prev, rn = nodes_[root].next;
        if (rn >= 0) nodes_[rn].prev = rp;
        if (rp >= 0) nodes_[rp].next = rn;
        if (first >= 0) nodes_[first].prev = root;
        nodes_[root].next = first;
        nodes_[root].prev = -1;
    }
    // TreeNode.split (resize of a tree bin): relink into lo / hi lists in `next` order; short lists become plain chains,
    // long ones are treeified again — unless the other list is empty, then the existing tree is kept as it is
    void split(int32_t b, int32_t index, int32_t bit) {
        int32_t loHead = -1, loTail = -1, hiHead = -1, hiTail = -1;
        int lc = 0, hc = 0;
        for (int32_t e = b, next; e >= 0; e = next) {
            next = nodes_[e].next;
            nodes_[e].next = -1;
            if ((nodes_[e].hash & bit) == 0) {
                nodes_[e].prev = loTail;
                if (loTail < 0) loHead = e; else nodes_[loTail].next = e;
                loTail = e; ++lc;
            } else {
                nodes_[e].prev = hiTail;
                if (hiTail < 0) hiHead = e; else nodes_[hiTail].next = e;
                hiTail = e; ++hc;
            }
        }
        if (loHead >= 0) {
            if (lc <= kUntreeifyThreshold) { untreeify(loHead); tab_[index] = loHead; }
            else { tab_[index] = loHead; if (hiHead >= 0) treeify(loHead); }
        }
        if (hiHead >= 0) {
            if (hc <= kUntreeifyThreshold) { untreeify(hiHead); tab_[index + bit] = hiHead; }
            else { tab_[index + bit] = hiHead; if (loHead >= 0) treeify(hiHead); }
        }
    }
    void untreeify(int32_t head) {
        for (int32_t q = head; q >= 0; q = nodes_[q].next) { Node& p = nodes_[q]; p.tree = false; p.prev = p.parent = p.left = p.right = -1; p.red = false; }
    }
    int32_t rotate_left(int32_t root, int32_t p) {
        int32_t r;
        if (p >= 0 && (r = nodes_[p].right) >= 0) {
            const int32_t rl = nodes_[p].right = nodes_[r].left;
            if (rl >= 0) nodes_[rl].parent = p;
            const int32_t pp = nodes_[r].parent = nodes_[p].parent;
            if (pp < 0) { root = r; nodes_[r].red = false; }
            else if (nodes_[pp].left == p) nodes_[pp].left = r;
            else nodes_[pp].right = r;
            nodes_[r].left = p;
            nodes_[p].parent = r;
        }
        return root;
    }
    int32_t rotate_right(int32_t root, int32_t p) {
        int32_t l;
        if (p >= 0 && (l = nodes_[p].left) >= 0) {
            const int32_t lr = nodes_[p].left = nodes_[l].right;
            if (lr >= 0) nodes_[lr].parent = p;
            const int32_t pp = nodes_[l].parent = nodes_[p].parent;
            if (pp < 0) { root = l; nodes_[l].red = false; }
            else if (nodes_[pp].right == p) nodes_[pp].right = l;
            else nodes_[pp].left = l;
            nodes_[l].right = p;
            nodes_[p].parent = l;
        }
        return root;
    }
    int32_t balance_insertion(int32_t root, int32_t x) {
        nodes_[x].red = true;
        for (int32_t xp, xpp, xppl, xppr;;) {
            if ((xp = nodes_[x].parent) < 0) { nodes_[x].red = false; return x; }
            if (!nodes_[xp].red || (xpp = nodes_[xp].parent) < 0) return root;
            if (xp == (xppl = nodes_[xpp].left)) {
                if ((xppr = nodes_[xpp].right) >= 0 && nodes_[xppr].red) {
                    nodes_[xppr].red = false; nodes_[xp].red = false; nodes_[xpp].red = true;
                    x = xpp;
                } else {
                    if (x == nodes_[xp].right) {
                        root = rotate_left(root, x = xp);
                        xpp = ((xp = nodes_[x].parent) < 0) ? -1 : nodes_[xp].parent;
                    }
                    if (xp >= 0) {
                        nodes_[xp].red = false;
                        if (xpp >= 0) { nodes_[xpp].red = true; root = rotate_right(root, xpp); }
                    }
                }
            } else {
                if (xppl >= 0 && nodes_[xppl].red) {
                    nodes_[xppl].red = false; nodes_[xp].red = false; nodes_[xpp].red = true;
                    x = xpp;
                } else {
                    if (x == nodes_[xp].left) {
                        root = rotate_right(root, x = xp);
                        xpp = ((xp = nodes_[x].parent) < 0) ? -1 : nodes_[xp].parent;
                    }
                    if (xp >= 0) {
                        nodes_[xp].red = false;
                        if (xpp >= 0) { nodes_[xpp].red = true; root = rotate_left(root, xpp); }
                    }
                }
            }
        }
    }
};

}  // namespace jdk
}  // namespace fspann
