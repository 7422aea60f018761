"""TEST INFRASTRUCTURE ONLY (oracle/): the ITERATION ORDER of libstdc++'s std::unordered_map<uint64_t, T> as a function of the sequence of keys inserted (operator[] /
emplace of absent keys).  The reference's deletion_wfa_po_poa (include/centrolign/alignment.hpp:2036-2282) iterates such a map when it picks the junction of its two
half-alignments, and strict '<' keeps the FIRST of equally good junctions: the oracle of that routine (oracle/wfa_oracle.py) needs the order.

What is emulated (GCC's _Hashtable with unique keys, identity hash for integers, max load factor 1): one singly linked list of all nodes; a bucket points at the node
BEFORE its first node.  A new node goes to the front of its bucket's run — or, if the bucket is empty, to the front of the whole list.  Before an insertion that would
exceed the bucket count the table grows to the next prime of libstdc++'s list at or above twice the count (13 at the first insertion) and relinks every node in list
order by the same rule.  Checked against the real container: tests/golden/std_unordered_order.json (made by tests/golden/make_std_unordered_order.py from a g++
program) and live where g++ is present (tests/test_wfa_oracle.py)."""

_PRIMES = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 103, 109, 113, 127, 137, 139, 149, 157, 167, 179, 193, 199, 211, 227,
           241, 257, 277, 293, 313, 337, 359, 383, 409, 439, 467, 503, 541, 577, 619, 661, 709, 761, 823, 887, 953, 1031, 1109, 1193, 1289, 1381, 1493, 1613, 1741, 1879,
           2029, 2179, 2357, 2549, 2753, 2971, 3209, 3469, 3739, 4027, 4349, 4703, 5087, 5503, 5953, 6427, 6949, 7517, 8123, 8783, 9497, 10273, 11113, 12011, 12983,
           14033, 15173, 16411, 17749, 19183, 20753, 22447, 24281, 26267, 28411, 30727, 33223, 35933, 38873, 42043, 45481, 49201, 53201, 57557, 62233, 67307, 72817,
           78779, 85229, 92203, 99733, 107897, 116731, 126271, 136607, 147793, 159871, 172933, 187091, 202409, 218971, 236897, 256279, 277261, 299951, 324503, 351061]


def _next_prime(n):
    for p in _PRIMES:
        if p >= n:
            return p
    raise ValueError("table larger than the emulation's prime list")


class UnorderedKeys:
    """keys of a std::unordered_map<uint64_t, T>: insert(key) (no effect if present), iteration in the container's order"""

    def __init__(self):
        self.n_buckets = 1
        self.next_resize = 0
        self.nxt = {}            # node -> next node (None at the end); the list's head is self.head
        self.head = None
        self.before = {}         # bucket -> the node before its first node ("HEAD" for the list's front)
        self.count = 0

    def __contains__(self, key):
        return key in self.nxt

    def _link(self, key, before, n_buckets):
        b = key % n_buckets
        if b in before:                              # front of the bucket's run
            prev = before[b]
            if prev == "HEAD":
                self.nxt[key], self.head = self.head, key
            else:
                self.nxt[key] = self.nxt[prev]
                self.nxt[prev] = key
        else:                                        # front of the whole list; the old first node's bucket now starts behind this node
            self.nxt[key] = self.head
            if self.head is not None:
                before[self.head % n_buckets] = key
            self.head = key
            before[b] = "HEAD"

    def insert(self, key):
        if key in self.nxt:
            return
        if self.count + 1 > self.next_resize:
            want = max(self.count + 1, 11 if self.next_resize == 0 else 0)
            if want >= self.n_buckets:
                new_n = _next_prime(max(want + 1, 2 * self.n_buckets))
                order = list(self)
                self.n_buckets, self.next_resize = new_n, new_n
                self.nxt, self.head, self.before = {}, None, {}
                for k in order:
                    self._link(k, self.before, new_n)
            else:
                self.next_resize = self.n_buckets
        self._link(key, self.before, self.n_buckets)
        self.count += 1

    def __iter__(self):
        k = self.head
        while k is not None:
            yield k
            k = self.nxt[k]
