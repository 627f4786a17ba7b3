"""Directed engine cases shared by the CPU (oracle vs ENGINE_SPEC) and GPU (HIP vs oracle) tests: crafted boards,
every piece x orientation x position, scripted clears."""
import numpy as np


def piece_cells(oracle):
    """cells[piece][rot] = [(dx, dy)] relative to the 4x4 box, read back from the oracle's own renderer."""
    out = {}
    for p in range(7):
        for r in range(4):
            g = oracle.Game(seed=0)
            g.g["rows"][0][:] = 0
            g.g["piece"], g.g["rot"], g.g["x"], g.g["y"] = p, r, 3, 8
            s = g.getState()
            out[(p, r)] = [(int(c) - 3, int(rw) - 8) for rw, c in np.argwhere(s == -1)]
    return out


def crafted_boards():
    """rows[20] (10-bit masks, bit c = column c): empty, 1-4 nearly full rows with a well, a staircase, walls of
    junk on both sides, a nearly topped-out stack, and a seeded random fill."""
    boards = {}
    boards["empty"] = np.zeros(20, np.uint16)
    for n in (1, 2, 3, 4):
        for well in (0, 4, 9):
            b = np.zeros(20, np.uint16)
            b[20 - n:] = 0x3FF & ~(1 << well)
            boards["full%d_well%d" % (n, well)] = b
    b = np.zeros(20, np.uint16)
    for c in range(10):
        for r in range(19 - c, 20):
            b[r] |= 1 << c
    b[19] &= 0x3FE
    boards["stairs"] = b
    b = np.zeros(20, np.uint16)
    b[6:] = 0b1100000011
    b[12:] |= 0b0010000100
    boards["walls"] = b
    b = np.zeros(20, np.uint16)
    b[3:] = 0x3FF & ~(0b11 << 4)
    boards["high_shaft"] = b
    rng = np.random.default_rng(5)
    b = np.zeros(20, np.uint16)
    for r in range(8, 20):
        m = int(rng.integers(0, 0x3FF))
        b[r] = m if m != 0x3FF else 0x3FE
    boards["random"] = b
    return boards


def valid(cells, rows, x, y):
    for dx, dy in cells:
        c, r = x + dx, y + dy
        if c < 0 or c > 9 or r < 0 or r > 19 or (int(rows[r]) >> c) & 1:
            return False
    return True


def sweep_states(oracle, app):
    """Packed games (oracle GAME_DTYPE records) for every board x piece x orientation x valid (x, y) x drop counter x
    back-to-back flag x combo, and the action list to apply to each."""
    from oracle.binding import GAME_DTYPE
    cells = piece_cells(oracle)
    recs = []
    for name, rows in crafted_boards().items():
        for p in range(7):
            for r in range(4):
                for x in range(-2, 9):
                    ys = [y for y in range(-1, 19) if valid(cells[(p, r)], rows, x, y)]
                    if not ys:
                        continue
                    # resting position(s), one above, and two high ones
                    rest = [y for y in ys if not valid(cells[(p, r)], rows, x, y + 1)]
                    pick = sorted(set(rest + [y - 1 for y in rest if y - 1 in ys] + ys[:2]))
                    for y in pick:
                        for dc in range(app):
                            g = np.zeros(1, GAME_DTYPE)
                            g["rows"][0] = rows
                            g["piece"], g["rot"], g["x"], g["y"] = p, r, x, y
                            g["drop_ctr"], g["flags"], g["combo"] = dc, (2 if (x + y) % 3 == 0 else 0), (-1 if (x + p) % 2 else 1)
                            g["piece_count"], g["seed"], g["score"], g["line_clears"] = 5 + p, 1000 + 7 * x + y, 1234, 3
                            recs.append(g)
    return np.concatenate(recs)
