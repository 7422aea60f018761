"""Makes tests/golden/std_unordered_order.json: insertion sequences into a real std::unordered_map<uint64_t, std::vector<int>> (g++'s libstdc++, the reference's
container in deletion_wfa_po_poa) and the iteration order / bucket count it ends with — the fixture oracle/std_unordered_order.py is checked against.
usage: python tests/golden/make_std_unordered_order.py   (needs g++)"""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
PROBE = r'''
#include <unordered_map>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
int main() {
    std::mt19937_64 rng(12345);
    const int ncase = 40;
    printf("[");
    for (int c = 0; c < ncase; ++c) {
        std::unordered_map<uint64_t, std::vector<int>> m;
        const int n = 1 + (int)(rng() % (c < 20 ? 60 : 1500));
        const uint64_t range = 1 + rng() % (c % 3 == 0 ? 50 : c % 3 == 1 ? 5000 : 1000000);
        printf("%s{\"keys\":[", c ? "," : "");
        for (int i = 0; i < n; ++i) { uint64_t k = rng() % range; m[k].push_back(i); printf("%s%llu", i ? "," : "", (unsigned long long)k); }
        printf("],\"order\":[");
        bool first = true;
        for (auto& kv : m) { printf("%s%llu", first ? "" : ",", (unsigned long long)kv.first); first = false; }
        printf("],\"buckets\":%zu}", m.bucket_count());
    }
    printf("]\n");
}
'''


def run():
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "probe.cpp"), os.path.join(d, "probe")
        with open(src, "w") as f:
            f.write(PROBE)
        subprocess.check_call(["g++", "-O1", "-std=c++11", src, "-o", exe])
        return subprocess.check_output([exe]).decode()


if __name__ == "__main__":
    with open(os.path.join(HERE, "std_unordered_order.json"), "w") as f:
        f.write(run())
