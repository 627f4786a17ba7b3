"""Functional model of the opt-in two-levels-per-round-trip walk (tree.hip, -DTM_WALK2): the register-set rotation,
the restart after a misprediction (outside the steady loop, always into rotation position 0) and the prediction repairs,
transliterated from the macros and run on random trees whose "best child" changes between walks.  A load returns a snapshot of the record at issue time; selection only ever
looks at the record / statistics of the slot that is current.  Checks: the visited path equals the plain walk's, the
slot that is current always holds the true node, and reports the restart rate.   python scripts/walk2_model.py"""
import random


def make_tree(rng, n_nodes):
    kids = {0: []}
    for i in range(1, n_nodes):
        kids[i] = []
    order = list(range(1, n_nodes))
    for i in order:
        if i * 3 + 1 < n_nodes and rng.random() < 0.9:
            k = rng.randint(1, 4)
            cs = [c for c in rng.sample(range(i + 1, min(n_nodes, i + 40)), k)]
            kids[i] = cs
    return kids


def main(seed=1, walks=3000, n_nodes=600):
    rng = random.Random(seed)
    kids = make_tree(rng, n_nodes)
    best = {i: (rng.choice(kids[i]) if kids[i] else 0) for i in kids}      # the "truth" of a selection at node i
    p = {i: 0 for i in kids}
    pp = {i: 0 for i in kids}
    levels = restarts = 0

    def load(i):           # record snapshot: (node, children, p, pp)
        return (i, tuple(kids[i]), p[i], pp[i])

    def stat(rec):         # statistics of rec's children: remembers WHOSE they are
        return rec[0]

    for w in range(walks):
        # a few selections change between walks
        for _ in range(3):
            i = rng.randrange(1, n_nodes)
            if kids[i]:
                best[i] = rng.choice(kids[i])
        # plain walk
        plain, x = [], 1
        while True:
            plain.append(x)
            if not kids[x]:
                break
            x = best[x]
        # pipelined walk
        r, q, s = [None] * 6, [0] * 6, [None] * 3

        def start(c, A, SA, B, C, SB, D):
            q[A] = c
            r[A] = load(c)
            q[B] = r[A][2]
            q[C] = r[A][3]
            s[SA] = stat(r[A])
            r[B] = load(q[B])
            r[C] = load(q[C])
            q[D] = r[B][3]
            s[SB] = stat(r[B])
            r[D] = load(q[D])
        path, prev_node, prev_pp = [], 0, 0
        restart, done = 1, False
        while not done:
            start(restart, 0, 0, 1, 2, 1, 3)           # TM_WALK_START(restart_node, r0, s0, r1, r2, s1, r3)
            j = 0
            while True:
                RC, S_C = j % 6, j % 3
                R1 = (j + 1) % 6
                R2, S2 = (j + 2) % 6, (j + 2) % 3
                R4 = (j + 4) % 6
                node, ch, my_p, my_pp = r[RC]
                assert node == q[RC]
                path.append(node)
                if not ch:
                    done = True
                    break
                q[R4] = r[R2][3]
                s[S2] = stat(r[R2])
                r[R4] = load(q[R4])
                assert s[S_C] == node, "statistics of another node in the current slot"
                c = best[node]
                levels += 1
                if my_p != c:
                    p[node] = c
                if prev_node and prev_pp != c:
                    pp[prev_node] = c
                prev_node, prev_pp = node, my_pp
                if c != q[R1]:
                    restart = c
                    restarts += 1
                    break
                j += 1
        assert path == plain, (w, path, plain)
    print("walks %d, levels %d, restarts per level %.4f" % (walks, levels, restarts / levels))


if __name__ == "__main__":
    main()
