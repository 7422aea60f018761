// chain_device.h — device-side layout of a packed chaining problem (shared by cl_chain_api.cpp and chain_kernels.hip)
#ifndef CL_CHAIN_DEVICE_H
#define CL_CHAIN_DEVICE_H

#include <stdint.h>

#define CL_CHAIN_NEG (-3.402823466e+38f)  // numeric_limits<float>::lowest(), the reference's mininf (anchorer.hpp:1868)

constexpr uint32_t kChainBlock = 256;     // match pairs per sequential block (one workgroup in the intra kernel)
constexpr uint32_t kChainFarTile = 1024;  // predecessor records per workgroup of a far launch
constexpr uint32_t kChainNearTile = 256;  // ... of a near launch, and of a far launch whose grid would not fill the chip
constexpr uint32_t kChainFullGrid = 1024; // workgroups that fill the chip (4 per CU)
constexpr uint32_t kChainFarGroup = 1;    // consecutive blocks served by one far launch
constexpr uint32_t kChainFarStreams = 2;  // auxiliary streams the far launches alternate between
constexpr uint32_t kChainLdsRecs = 1024;  // records of one start-node group broadcast through LDS (more fall back to HBM)
constexpr uint32_t kChainMacro = 1024;    // match pairs per launch of the walk kernel (one workgroup per chain combination)
constexpr uint32_t kChainWalkMaxCombos = 256; // the walk's workgroups wait for one another: all of them must be resident (one per CU: the affine
                                              // walk takes 96 VGPRs at 1 024 threads)
constexpr uint32_t kChainWalkSweepCombos = 16; // up to here every workgroup reads every other's granule; beyond, one atomic maximum + arrival count per pair
                                               // (10 x 1 Mbp, device time of a merge's DPs, granules / reduction: 4 combinations 480 / 502 ms, 25 combinations 1 091 / 1 052 ms)

struct ClChainParams {
    double gap_open[3];
    double gap_extend[3];
    double scale;         // local_scale
};

// one (chain1, chain2) combination = one set of the reference's search trees (anchorer.hpp:2087-2237)
struct ClChainCombo {
    uint32_t n_recs;
    // records: the match pairs that lie on this chain pair, in sorted pair order
    const uint32_t* rec_s;    // sorted pair index
    const uint32_t* ins_t;    // index_on(e1, p1)            : insertion time on chain p1
    const uint32_t* off;      // index_on(e2, p2)            : key offset (anchorer.hpp:1894-1896)
    const int32_t*  sigma;    // source shift                 (:1875-1880)
    float*          val;      // [7][n_recs]: dp, then the value stored in tree pw = 0..5 (:2325-2342)
    const uint32_t* prefix;   // [n_blocks + 1] records whose pair index is < block start
    // queries: one per sorted pair (dense)
    const uint32_t* qt;       // predecessor_index(b1, p1), 0xFFFFFFFF = no forward edge on p1
    const uint32_t* qoff;     // predecessor_index(b2, p2) + 1 (wraps to 0)   (:1898-1901)
    const int32_t*  q;        // query shift                  (:1886-1892)
    int*            acc;      // [n_pairs][7] running maxima, order-preserving float encoding
    uint32_t*       own_rec;  // [n_pairs] position of the pair's record in this combination, 0xFFFFFFFF = none
};

struct ClChainDevice {
    uint32_t n_pairs, n_combos;
    const ClChainCombo* combos;
    const float* weight;        // anchor weight per sorted pair
    const float* init;          // value of the chain that starts at the pair (weight, + lead indel / -inf under global anchoring)
    float* dp;                  // final DP value per sorted pair
    const uint32_t* rec_off;    // [n_pairs + 1] records of each pair
    const uint32_t* rec_combo;
    const uint32_t* rec_pos;
    const uint32_t* group;      // [n_pairs] group id, consecutive along the sorted pairs; pairs of one group cannot precede
                                // one another
    const uint32_t* grp_base;   // [n_pairs] LDS slot of the pair's first record within its (block, group)
    const uint32_t* grp_total;  // [n_pairs] number of records of the pair's (block, group)
    ClChainParams params;
    uint32_t sparse;            // sparse_chain_dp: only the gap-free maximum (acc[0], val[0]) is used
    const uint32_t* group_end;  // [n_pairs] sorted index one past the last pair of the pair's group
    unsigned long long* xch;    // [n_combos][kChainMacro] {tag, value} granules: the walk's workgroups exchange their
                                // combination's best candidate per pair (zeroed before every DP)
    uint32_t* xred;             // null, or [n_pairs][2] {biased maximum, arrivals}: the exchange as a reduction (many combinations; zeroed before every DP)
    unsigned long long* xdp;    // chain_walk2.hip: [n_combos][kChainMacro] {tag, DP value} granules a combination's main workgroup leaves for its helpers, and
    unsigned long long* hacc;   // [n_combos][kChainMacro][8] {tag, maximum} granules the helpers give back (both zeroed before every DP; null: chain_walk_kernel)
    uint32_t debug;             // chain_walk2.hip: count and clock into status[8..23] (CL_CHAIN_WALK2_DEBUG)
    uint32_t* status;           // [1] set non-zero by a walk that gave up waiting for a sibling workgroup
    // branch-and-bound far pass (chain_far.hip); null / 0 when it is not in use
    int* far_rec;               // [r_pad][12] the records of every combination as the kernels consume them: insertion index, offset,
                                // shift, 7 encoded stored values (written by the walk), 2 pad words; combination c starts at far_base[c]
    const uint32_t* far_base;
    uint32_t lo_mask;           // the first source record of an inter launch is rounded down to a multiple of lo_mask + 1
    // status[2..3], status[4..5]: 64-bit counts of leaves the branch-and-bound far pass scanned / had in range
};

// ---- branch-and-bound far pass ---------------------------------------------------------------------------------------
// The records of a combination, in sorted-pair order, are cut into aligned nodes of 64 * 8^l records (l = 0 .. n_levels-1).
// Once every record of a node is final the node is SEALED: in two static orders of its records (by offset; by shift bucket,
// then offset) the running maximum of the DP value is written down.  A query can then bound, by a few binary searches, the
// best candidate the node could give it, and skip the node when that is below what it already has.
constexpr int kFarLeafShift = 6;
constexpr int kFarFanShift = 3;
constexpr int kFarMaxLevels = 4;     // up to 32768 records: one wave seals a node in about a millisecond
constexpr uint32_t kFarLag = 8;       // upper limit of the far pass's lag: the far pass of macro-block k reads the records final `lag`
                                      // macro-blocks earlier, so that many far launches run side by side (cl_chain_api.cpp picks the lag)
constexpr int kFarBandShift = 16;   // shift buckets of 65536 by default (ClFarDevice::band_shift; wider when the shifts span more than 2^15 buckets): beyond
                                    // that the gap cost is on its last, nearly flat piece.  Measured on 2 x 1 Mbp: buckets of 4096 / 1024 / 256 open 3.4x /
                                    // 3.6x / 3.7x the leaves (the "everything else pays one bucket width" term gets weak), 572 / 751 / 812 ms against 366

// Every level keeps its records' keys in two static orders per node (by offset; by shift bucket, then offset), each as a static
// 8-ary search tree (an "S+ tree"): the ascending keys lie in blocks of eight, every block followed by the eight RUNNING MAXIMA of the DP
// value at its positions (written when the node is sealed), and index arrays hold every 8th, 64th, ... key.  A search reads one 32-byte
// index block per tree level and ends on one 64-byte block that holds both the last key below the bound and the running maximum there:
// level + 2 dependent loads per node instead of the 6 .. 15 + 2 of a binary search.  All arrays live in ONE arena; the kernels keep the
// table of their offsets in LDS (a lane picks its node's level at run time).
constexpr int kFarTabWidth = 2 + 2 * kFarMaxLevels;   // words per level in ClFarDevice::tab
constexpr uint32_t kPeerMaxMembers = 8;    // contexts that share the far pass of one merge
constexpr uint32_t kPeerMaxCombos = 64;    // chain combinations of a DP whose far pass is shared (an inbox slot holds them all)
constexpr uint32_t kPeerRing = 32;         // inbox slots: a member is never more than 2 * lag + 2 <= 18 macro-blocks ahead of another
constexpr uint32_t kPeerSlotInts = kPeerMaxCombos * 1024u * 7u;   // (kChainMacro pairs per macro-block)
// the words behind the inbox slots of an exported allocation (cl_peer_api.cpp), in uint32 units from its `flags` pointer
constexpr uint32_t kPeerFlagWords = kPeerMaxMembers * kPeerRing;   // arrival words [member][slot]
constexpr uint32_t kPeerTestWords = kPeerMaxMembers;               // cl_context_peer_selftest: one arrival word per member
constexpr uint32_t kPeerTestInts = 8 * kPeerMaxMembers;            // ... and where the members' kernels store
constexpr uint32_t kPeerStealWords = 4;                            // cl_context_peer_steal: the 64-bit counter (member 0's is the group's) + where a steal's answer lands
constexpr uint32_t kPeerDoneWords = kPeerMaxMembers;               // "member m has folded every slot of shared DP e": the next shared DP's peer stores wait for it
constexpr uint32_t kPeerStealAt = kPeerFlagWords + kPeerTestWords + kPeerTestInts;
constexpr uint32_t kPeerDoneAt = kPeerStealAt + kPeerStealWords;
constexpr uint32_t kPeerTailWords = kPeerDoneAt + kPeerDoneWords;

struct ClFarDevice {
    uint32_t n_levels, r_pad;
    uint32_t* arena;             // all search structures of all levels
    // per level, in BLOCKS OF 8 WORDS from `arena` (far_at, chain_far.hip: 2^35 words): [0] the blocked offset order (2 * r_pad words: 8 keys | 8 running maxima per block), [1] the blocked
    // (bucket << off_bits | offset) order (0xFFFFFFFF: sparse_chain_dp has none), [2 + j] every 8^(j+1)-th key of the offset order,
    // [2 + kFarMaxLevels + j] the same for the bucket order  (j = 0 .. level)
    uint32_t tab[kFarMaxLevels][kFarTabWidth];
    const uint32_t* tab_dev;     // the same table in device memory (the far kernel copies it to LDS)
    uint32_t* perm_o[kFarMaxLevels];   // [r_pad] which record sits at each position of the two orders (the sealing kernel's gather)
    uint32_t* perm_b[kFarMaxLevels];
    int32_t sig_bias;            // added to a shift before bucketing (buckets are positive)
    uint32_t band_shift;         // log2 of the bucket width
    uint32_t off_bits;           // bucket keys are bucket << off_bits | offset
    double band_pen;             // least gap cost of a shift difference of one bucket width or more
    double slack_t0;             // rounding allowance of a bound: 2^-21 (|dp| + slack_t0 + slack_e0 |query shift| + |weight|)
    double slack_e0;
    // far pass shared by the contexts of a merge group (one context per process / GPU, every member runs the same DP; cl_peer_api.cpp):
    // this member bounds the queries of the combinations c = share_i + j * share_n only and stores what it finds — every query of those
    // combinations, found or not — into the other members' inboxes as well ([combination][pair of the macro-block][7] encoded maxima)
    uint32_t share_n, share_i;   // share_n <= 1: no sharing
    int* peer_out[kPeerMaxMembers - 1];   // the block's slot in each other member's inbox (null: none)
    // XCD-aware far launch (CL_CHAIN_FAR_XCD=1): a 1-D grid whose workgroup w — dispatched to XCD w % 8 — takes a combination = w % 8 (mod 8), so that an XCD's L2 holds the
    // search structures of every eighth combination instead of all of them.  xcd_x: workgroups per combination (0: the 2-D grid of rounds 2-4), xcd_n: combinations of the launch
    uint32_t xcd_x, xcd_n;
};

#endif
