"""Test infrastructure: a THIRD, independent restatement of java.util.HashMap (JDK 21) with tree bins, in plain Python objects —
written to read like the JDK source (Node / TreeNode objects with next / prev / parent / left / right / red), so that the two
C++ models (oracle/fspann_oracle.cpp: JHashMap; fspann-query-system_amd/host/java_hashmap.hpp: the product's rare path) can be
compared with something that shares no code with either.  Keys are ints standing for Strings: `hash_of(key)` is
String.hashCode, `compare(a, b)` is String.compareTo (only consulted for different keys with equal hashCode).

No JVM exists in the build container: this pins coding slips, not the recollection of the JDK itself (parity stays unpinned
until java/com/fspann/gpu/GoldenDumper.java reproduces tests/golden on a JVM)."""

TREEIFY_THRESHOLD, UNTREEIFY_THRESHOLD, MIN_TREEIFY_CAPACITY = 8, 6, 64


def _i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def spread(h):
    u = h & 0xFFFFFFFF
    return _i32(u ^ (u >> 16))


def table_size_for(cap):
    c = (cap - 1) & 0xFFFFFFFF
    nlz = 32 if c == 0 else 32 - c.bit_length()
    n = _i32(0xFFFFFFFF >> (nlz & 31))
    return 1 if n < 0 else ((1 << 30) if n >= (1 << 30) else n + 1)


class Node:
    __slots__ = ("hash", "key", "value", "next", "prev", "parent", "left", "right", "red", "tree")

    def __init__(self, h, k, v, nxt):
        self.hash, self.key, self.value, self.next = h, k, v, nxt
        self.prev = self.parent = self.left = self.right = None
        self.red = False
        self.tree = False


class JavaHashMap:
    def __init__(self, initial_capacity, hash_of, compare):
        self.table = None
        self.threshold = table_size_for(max(0, initial_capacity))
        self.size = 0
        self.hash_of, self.compare = hash_of, compare
        self.treeified = False
        self.unmodelled = False

    # ---- HashMap --------------------------------------------------------------------------------------------------
    def put(self, key, value):
        h = spread(self.hash_of(key))
        if not self.table:
            self.resize()
        tab = self.table
        n = len(tab)
        i = (n - 1) & h
        p = tab[i]
        if p is None:
            tab[i] = Node(h, key, value, None)
        else:
            e = None
            if p.hash == h and p.key == key:
                e = p
            elif p.tree:
                e = self.put_tree_val(p, h, key, value)
            else:
                bin_count = 0
                while True:
                    e = p.next
                    if e is None:
                        p.next = Node(h, key, value, None)
                        if bin_count >= TREEIFY_THRESHOLD - 1:
                            self.treeify_bin(h)
                        break
                    if e.hash == h and e.key == key:
                        break
                    p = e
                    bin_count += 1
            if e is not None:
                e.value = value
                return False
        self.size += 1
        if self.size > self.threshold:
            self.resize()
        return True

    def get(self, key):
        if not self.table:
            return None
        h = spread(self.hash_of(key))
        e = self.table[(len(self.table) - 1) & h]
        while e is not None:                       # the `next` list holds every node of the bin, tree or not
            if e.hash == h and e.key == key:
                return e.value
            e = e.next
        return None

    def resize(self):
        old = self.table
        old_cap = len(old) if old else 0
        old_thr = self.threshold
        new_thr = 0
        if old_cap > 0:
            new_cap = old_cap << 1
            if old_cap >= 16:
                new_thr = old_thr << 1
        elif old_thr > 0:
            new_cap = old_thr
        else:
            new_cap, new_thr = 16, 12
        if new_thr == 0:
            new_thr = int(new_cap * 0.75)
        self.threshold = new_thr
        self.table = tab = [None] * new_cap
        if old:
            for j in range(old_cap):
                e = old[j]
                if e is None:
                    continue
                if e.next is None:
                    tab[e.hash & (new_cap - 1)] = e
                elif e.tree:
                    self.split(e, tab, j, old_cap)
                else:
                    lo_head = lo_tail = hi_head = hi_tail = None
                    while e is not None:
                        nxt = e.next
                        if (e.hash & old_cap) == 0:
                            if lo_tail is None:
                                lo_head = e
                            else:
                                lo_tail.next = e
                            lo_tail = e
                        else:
                            if hi_tail is None:
                                hi_head = e
                            else:
                                hi_tail.next = e
                            hi_tail = e
                        e = nxt
                    if lo_tail is not None:
                        lo_tail.next = None
                        tab[j] = lo_head
                    if hi_tail is not None:
                        hi_tail.next = None
                        tab[j + old_cap] = hi_head

    def treeify_bin(self, h):
        tab = self.table
        n = len(tab)
        if n < MIN_TREEIFY_CAPACITY:
            self.resize()
            return
        index = (n - 1) & h
        e = tab[index]
        if e is None:
            return
        hd = tl = None
        while e is not None:
            e.tree = True
            e.parent = e.left = e.right = None
            e.red = False
            e.prev = tl
            if tl is None:
                hd = e
            tl = e
            e = e.next
        tab[index] = hd
        self.treeified = True
        self.treeify(hd, tab)

    def items(self):
        for first in (self.table or []):
            e = first
            while e is not None:
                yield e.key, e.value
                e = e.next

    # ---- TreeNode ---------------------------------------------------------------------------------------------------
    def _dir(self, h, k, p):
        if p.hash > h:
            return -1
        if p.hash < h:
            return 1
        c = self.compare(k, p.key)
        if c == 0:                                   # tieBreakOrder(identityHashCode): not reproducible outside a JVM
            self.unmodelled = True
            return -1
        return -1 if c < 0 else 1

    def move_root_to_front(self, tab, root):
        if root is None or not tab:
            return
        index = (len(tab) - 1) & root.hash
        first = tab[index]
        if root is not first:
            tab[index] = root
            rp, rn = root.prev, root.next
            if rn is not None:
                rn.prev = rp
            if rp is not None:
                rp.next = rn
            if first is not None:
                first.prev = root
            root.next = first
            root.prev = None

    def treeify(self, head, tab):
        root = None
        x = head
        while x is not None:
            nxt = x.next
            x.left = x.right = None
            if root is None:
                x.parent = None
                x.red = False
                root = x
            else:
                p = root
                while True:
                    d = self._dir(x.hash, x.key, p)
                    xp = p
                    p = p.left if d <= 0 else p.right
                    if p is None:
                        x.parent = xp
                        if d <= 0:
                            xp.left = x
                        else:
                            xp.right = x
                        root = self.balance_insertion(root, x)
                        break
            x = nxt
        self.move_root_to_front(tab, root)

    def put_tree_val(self, first, h, k, v):
        root = first
        while root.parent is not None:
            root = root.parent
        p = root
        while True:
            if p.hash == h and p.key == k:
                return p
            d = self._dir(h, k, p)
            xp = p
            p = p.left if d <= 0 else p.right
            if p is None:
                xpn = xp.next
                x = Node(h, k, v, xpn)
                x.tree = True
                if d <= 0:
                    xp.left = x
                else:
                    xp.right = x
                xp.next = x
                x.parent = x.prev = xp
                if xpn is not None:
                    xpn.prev = x
                self.move_root_to_front(self.table, self.balance_insertion(root, x))
                return None

    def split(self, b, tab, index, bit):
        lo_head = lo_tail = hi_head = hi_tail = None
        lc = hc = 0
        e = b
        while e is not None:
            nxt = e.next
            e.next = None
            if (e.hash & bit) == 0:
                e.prev = lo_tail
                if lo_tail is None:
                    lo_head = e
                else:
                    lo_tail.next = e
                lo_tail = e
                lc += 1
            else:
                e.prev = hi_tail
                if hi_tail is None:
                    hi_head = e
                else:
                    hi_tail.next = e
                hi_tail = e
                hc += 1
            e = nxt
        if lo_head is not None:
            if lc <= UNTREEIFY_THRESHOLD:
                tab[index] = self.untreeify(lo_head)
            else:
                tab[index] = lo_head
                if hi_head is not None:
                    self.treeify(lo_head, tab)
        if hi_head is not None:
            if hc <= UNTREEIFY_THRESHOLD:
                tab[index + bit] = self.untreeify(hi_head)
            else:
                tab[index + bit] = hi_head
                if lo_head is not None:
                    self.treeify(hi_head, tab)

    @staticmethod
    def untreeify(head):
        q = head
        while q is not None:
            q.tree = False
            q.prev = q.parent = q.left = q.right = None
            q.red = False
            q = q.next
        return head

    @staticmethod
    def rotate_left(root, p):
        if p is not None and p.right is not None:
            r = p.right
            rl = p.right = r.left
            if rl is not None:
                rl.parent = p
            pp = r.parent = p.parent
            if pp is None:
                root = r
                r.red = False
            elif pp.left is p:
                pp.left = r
            else:
                pp.right = r
            r.left = p
            p.parent = r
        return root

    @staticmethod
    def rotate_right(root, p):
        if p is not None and p.left is not None:
            l = p.left
            lr = p.left = l.right
            if lr is not None:
                lr.parent = p
            pp = l.parent = p.parent
            if pp is None:
                root = l
                l.red = False
            elif pp.right is p:
                pp.right = l
            else:
                pp.left = l
            l.right = p
            p.parent = l
        return root

    def balance_insertion(self, root, x):
        x.red = True
        while True:
            xp = x.parent
            if xp is None:
                x.red = False
                return x
            if not xp.red:
                return root
            xpp = xp.parent
            if xpp is None:
                return root
            xppl = xpp.left
            if xp is xppl:
                xppr = xpp.right
                if xppr is not None and xppr.red:
                    xppr.red = False
                    xp.red = False
                    xpp.red = True
                    x = xpp
                else:
                    if x is xp.right:
                        x = xp
                        root = self.rotate_left(root, x)
                        xp = x.parent
                        xpp = None if xp is None else xp.parent
                    if xp is not None:
                        xp.red = False
                        if xpp is not None:
                            xpp.red = True
                            root = self.rotate_right(root, xpp)
            else:
                if xppl is not None and xppl.red:
                    xppl.red = False
                    xp.red = False
                    xpp.red = True
                    x = xpp
                else:
                    if x is xp.left:
                        x = xp
                        root = self.rotate_right(root, x)
                        xp = x.parent
                        xpp = None if xp is None else xp.parent
                    if xp is not None:
                        xp.red = False
                        if xpp is not None:
                            xpp.red = True
                            root = self.rotate_left(root, xpp)


def decimal_compare(a, b):
    """String.compareTo(Long.toString(a), Long.toString(b))"""
    sa, sb = str(a), str(b)
    for ca, cb in zip(sa, sb):
        if ca != cb:
            return ord(ca) - ord(cb)
    return len(sa) - len(sb)
