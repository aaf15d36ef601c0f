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
        {
            StringBuilder sb = new StringBuilder("cast");
            for (double x : new double[]{1e300, -1e300, Double.NaN, 2147483647.5, -2147483648.5, -0.5, 3.99})
                sb.append(' ').append((int) Math.floor(x));
            System.out.println(sb);
        }
    }
}
