package com.fspann.gpu;

import com.fspann.index.paper.Coding;
import com.fspann.index.paper.GreedyPartitioner;

import java.util.*;

/**
 * Pins the CPU oracle (oracle/fspann_oracle.cpp) against a real JVM — the missing step behind "parity unpinned".
 *
 * Not runnable in the build container (no JDK).  On any box with JDK 21 and the reference on the classpath:
 *
 *     javac -cp fsp-anns-parent/index/target/classes -d out java/com/fspann/gpu/GoldenDumper.java
 *     java  -cp out:fsp-anns-parent/index/target/classes com.fspann.gpu.GoldenDumper > jvm_golden.txt
 *     python tests/golden/compare_jvm_dump.py jvm_golden.txt        # compares with the oracle, line by line
 *
 * Every line is "tag value..." so the comparison is a plain text diff.  It covers exactly the JDK / reference
 * behaviours the oracle models from memory (SURVEY §9 Appendix A):
 *   splitmix   java.util.SplittableRandom stream + nextDouble
 *   gauss      Math.log / Math.cos Box-Muller values (HotSpot intrinsics: expected to match glibc to <= 1 ulp, not bitwise)
 *   quick      Coding.buildRandomG / H / C for the CodingQuickCheck input
 *   hashcode   String.hashCode of decimal ordinals
 *   hashmap    iteration order of HashMap<String,Long>(cap) after a put sequence (incl. resize and value updates)
 *   hashtree   the same with ids crowded into one bin: treeifyBin, putTreeVal, split / untreeify on resize
 *   pq         PriorityQueue<long[]>(comparingLong(a -> a[1])) poll order with ties
 *   key        GreedyPartitioner.computeKey / hamming on given BitSets
 *   build      GreedyPartitioner.build on a small id->code map (partition ids, min/max keys, representative)
 *   cast       (int) Math.floor(x) saturation
 *   route      ONE WHOLE lookupCandidatesWithScores list per line: partitions cut by GreedyPartitioner.build from a fixed id -> code
 *              map (three tables of 24 000 decimal ids), then the literal loop of PartitionedIndexService.java:619-696 — HashMap
 *              bestScore, PriorityQueue probing, HARD_CAP checks per probe step, stable sort by score — for two fixed queries under
 *              four (HARD_CAP, probes) settings: one query's bestScore map TREEIFIES a bin (twelve ids of one bin of the 2 048-table,
 *              on two score levels), two settings CROSS the HARD_CAP (one with a resize 128 -> 256 on the way).  No RocksDB, no
 *              crypto: metadata.isDeleted is "nothing deleted".  This pins the ORDER the kernels derive in closed form, not just
 *              the primitives behind it.
 */
public final class GoldenDumper {
    public static void main(String[] args) {
        for (long seed : new long[]{0L, 13L, 42L, 12345L}) {
            SplittableRandom r = new SplittableRandom(seed);
            StringBuilder sb = new StringBuilder("splitmix " + seed);
            for (int i = 0; i < 4; i++) sb.append(' ').append(Long.toHexString(r.nextLong()));
            SplittableRandom r2 = new SplittableRandom(seed);
            for (int i = 0; i < 2; i++) sb.append(' ').append(Long.toHexString(Double.doubleToLongBits(r2.nextDouble())));
            System.out.println(sb);
        }
        {   // CodingQuickCheck input (index/src/test/java/com/fspann/index/CodingQuickCheck.java:10-37)
            double[] v = new double[128];
            for (int i = 0; i < 128; i++) v[i] = i * 0.01;
            Coding.GFunction G = Coding.buildRandomG(128, 24, 2, 1.0, 12345L);
            StringBuilder sb = new StringBuilder("gauss");
            for (int i = 0; i < 8; i++) sb.append(' ').append(Long.toHexString(Double.doubleToLongBits(G.alpha[0][i])));
            System.out.println(sb);
            sb = new StringBuilder("quick_r");
            for (int j = 0; j < 24; j++) sb.append(' ').append(Long.toHexString(Double.doubleToLongBits(G.r[j])));
            System.out.println(sb);
            int[] H = Coding.H(v, G);
            System.out.println("quick_H " + Arrays.toString(H).replaceAll("[\\[\\],]", ""));
            System.out.println("quick_C " + Arrays.toString(Coding.C(v, G).toLongArray()).replaceAll("[\\[\\],]", ""));
        }
        {
            StringBuilder sb = new StringBuilder("hashcode");
            for (long o : new long[]{0, 9, 10, 999, 1000, 123456, 999999, 1000000}) sb.append(' ').append(Long.toString(o).hashCode());
            System.out.println(sb);
        }
        for (int cap : new int[]{4, 16, 64, 32768}) {   // put 0..199 (stride 37) then update a few values
            Map<String, Long> m = new HashMap<>(cap);
            for (int i = 0; i < 200; i++) m.put(Long.toString((i * 37L) % 1009), (long) i);
            m.put("37", -1L);
            StringBuilder sb = new StringBuilder("hashmap " + cap);
            for (String k : m.keySet()) sb.append(' ').append(k);
            System.out.println(sb);
        }
        // Tree bins (HashMap.TreeNode): decimal ids crowded into one bin of a 64-slot table -> treeifyBin at the ninth, putTreeVal
        // after it, a resize to 128 in the middle (split: halves stay trees or untreeify).  The iteration order of such a map is
        // what the library's host model (host/java_hashmap.hpp) and the oracle's JHashMap must reproduce.
        for (int bin : new int[]{5, 41}) {
            Map<String, Long> m = new HashMap<>(64);
            int crowded = 0;
            for (int i = 0; crowded < 40; i++) {
                int h = String.valueOf(i).hashCode();
                if (((h ^ (h >>> 16)) & 63) == bin) { m.put(String.valueOf(i), (long) i); crowded++; }
                if (i % 97 == 0) m.put(String.valueOf(1000003 + i), 0L);          // spread-out ids in between: the map passes 48 entries
            }
            StringBuilder sb = new StringBuilder("hashtree " + bin);
            for (String k : m.keySet()) sb.append(' ').append(k);
            System.out.println(sb);
        }
        {
            long[][] traces = {{5, -1, 3, 3, -1, -1}, {5, -1, 3, 4, -1, 4, -1, -1}, {5, -1, 3, 4, -1, 2, -1, -1}, {7, 7, 7, -1, 7, -1, -1, -1}};
            for (long[] ops : traces) {
                PriorityQueue<long[]> pq = new PriorityQueue<>(Comparator.comparingLong(a -> a[1]));
                StringBuilder sb = new StringBuilder("pq");
                for (int i = 0; i < ops.length; i++) {
                    if (ops[i] >= 0) pq.add(new long[]{i, ops[i]});
                    else if (!pq.isEmpty()) sb.append(' ').append(pq.poll()[0]);
                }
                System.out.println(sb);
            }
        }
        {
            BitSet a = BitSet.valueOf(new long[]{0b1011L}), b = BitSet.valueOf(new long[]{1L << 63, 1L});
            System.out.println("key " + GreedyPartitioner.computeKey(a) + " " + GreedyPartitioner.computeKey(b) + " " +
                    GreedyPartitioner.hamming(BitSet.valueOf(new long[]{0xFFL, 1L}), BitSet.valueOf(new long[]{0x0FL})));
        }
        {   // 150 ids, codes = 8 low bits of (i * 73) -> many equal keys: block membership depends on HashMap order
            Map<String, BitSet> idToCode = new HashMap<>(150);
            for (int i = 0; i < 150; i++) idToCode.put(Integer.toString(i), BitSet.valueOf(new long[]{(i * 73L) & 0x3FL}));
            List<GreedyPartitioner.Partition> parts = GreedyPartitioner.build(idToCode, 64);
            for (GreedyPartitioner.Partition p : parts)
                System.out.println("build " + p.minKey + " " + p.maxKey + " " + Arrays.toString(p.repCode.toLongArray()).replaceAll("[\\[\\],]", "")
                        + " " + String.join(",", p.ids));
        }
        {   // route: see the class comment.  Scene = tests/golden/compare_jvm_dump.py route_scene(), number for number.
            final int N = 24000, T = 3;
            final long[] A = {7919L, 104729L, 1299709L}, B = {17L, 4242L, 31337L};
            final long[] QA = {0x1234L, 0x0F0FL, 0x5555L}, QB = {0x8001L, 0x7FFEL, 0x00FFL};
            long[][] code = new long[T][N];
            for (int t = 0; t < T; t++)
                for (int i = 0; i < N; i++) code[t][i] = (((long) i * A[t] + B[t]) % 65521L) & 0xFFFFL;
            int crowded = 0;                          // the first twelve ids whose bin of a 2 048-slot table is 312: six beside QA in table 0,
            for (int i = 0; i < N && crowded < 12; i++) {   //   six one bit away from QA in table 1 (another partition distance)
                int h = Integer.toString(i).hashCode();
                if (((h ^ (h >>> 16)) & 2047) == 312) {
                    if (crowded < 6) code[0][i] = QA[0]; else code[1][i] = QA[1] ^ 0x8000L;
                    crowded++;
                }
            }
            List<List<GreedyPartitioner.Partition>> tables = new ArrayList<>();
            for (int t = 0; t < T; t++) {
                Map<String, BitSet> idToCode = new HashMap<>(N);
                for (int i = 0; i < N; i++) idToCode.put(Integer.toString(i), BitSet.valueOf(new long[]{code[t][i]}));
                tables.add(GreedyPartitioner.build(idToCode, 64));
            }
            final int[][] settings = {{1500, 5}, {100, 5}, {1500, 10}, {300, 5}};       // {HARD_CAP, perDivisionMaxProbes}
            for (int[] st : settings) {
                final int HARD_CAP = st[0], perDivisionMaxProbes = st[1];
                for (int qn = 0; qn < 2; qn++) {
                    final long[] q = qn == 0 ? QA : QB;
                    // ---- PartitionedIndexService.java:619-687, literally (one division per table; nothing deleted) ----
                    Map<String, Long> bestScore = new HashMap<>(Math.min(HARD_CAP, 1 << 16));
                    int rawSeen = 0;
                    for (int t = 0; t < T && bestScore.size() < HARD_CAP; t++) {
                        List<GreedyPartitioner.Partition> parts = tables.get(t);
                        BitSet qBits = BitSet.valueOf(new long[]{q[t]});
                        long qKey = GreedyPartitioner.computeKey(qBits);
                        int center = GreedyPartitioner.findNearestPartition(parts, qKey);
                        PriorityQueue<long[]> probeQueue = new PriorityQueue<>(Comparator.comparingLong(a -> a[1]));
                        boolean[] visited = new boolean[parts.size()];
                        long centerDist = GreedyPartitioner.hamming(qBits, parts.get(center).repCode);
                        probeQueue.add(new long[]{center, centerDist});
                        visited[center] = true;
                        int probesUsed = 0;
                        while (!probeQueue.isEmpty() && probesUsed < perDivisionMaxProbes && bestScore.size() < HARD_CAP) {
                            long[] cur = probeQueue.poll();
                            int idx = (int) cur[0];
                            probesUsed++;
                            GreedyPartitioner.Partition p = parts.get(idx);            // collectPartitionOrdered (PIS:726-753)
                            long partDist = GreedyPartitioner.hamming(qBits, p.repCode);
                            for (String id : p.ids) {
                                Long prev = bestScore.get(id);
                                if (prev == null || partDist < prev) { bestScore.put(id, partDist); rawSeen++; }
                            }
                            int left = idx - 1;
                            if (left >= 0 && !visited[left]) {
                                visited[left] = true;
                                probeQueue.add(new long[]{left, GreedyPartitioner.hamming(qBits, parts.get(left).repCode)});
                            }
                            int right = idx + 1;
                            if (right < parts.size() && !visited[right]) {
                                visited[right] = true;
                                probeQueue.add(new long[]{right, GreedyPartitioner.hamming(qBits, parts.get(right).repCode)});
                            }
                        }
                    }
                    // ---- PIS:690-696: the map's entries in iteration order, stable-sorted by score ----
                    List<Map.Entry<String, Long>> result = new ArrayList<>(bestScore.entrySet());
                    result.sort(Comparator.comparingLong(Map.Entry::getValue));
                    StringBuilder sb = new StringBuilder("route " + HARD_CAP + " " + perDivisionMaxProbes + " " + (qn == 0 ? "QA" : "QB") + " " + result.size() + " " + rawSeen);
                    for (Map.Entry<String, Long> e : result) sb.append(' ').append(e.getKey()).append(':').append(e.getValue());
                    System.out.println(sb);
                }
            }
        }
        {
            StringBuilder sb = new StringBuilder("cast");
            for (double x : new double[]{1e300, -1e300, Double.NaN, 2147483647.5, -2147483648.5, -0.5, 3.99})
                sb.append(' ').append((int) Math.floor(x));
            System.out.println(sb);
        }
    }
}
