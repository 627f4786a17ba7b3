#!/usr/bin/env python
"""play.py — self-play driver with the command line of the reference's play.py (flags: play.py:46-70), running
on the MI355X engine.  `python play.py --agent_type ValueSimLP --mcts_sims 500 --ngames 10` behaves like the
reference (same per-game log lines, play.py:25-37,164); `--n_games N` plays N games concurrently on the GPU
(new, optional).  Only the hot path is reimplemented: --save/--save_tree (PyTables episode files, util/Data.py),
--gui and --interactive belong to the reference's storage / UI layers and are refused with a clear message.
"""
import argparse
import sys

import numpy as np


def build_parser():
    p = argparse.ArgumentParser()
    add = p.add_argument
    add('--agent_type', default=None, type=str, help='Which agent to use')
    add('--app', default=1, type=int, help='Actions-per-drop')
    add('--benchmark', default=False, help='Benchmark mode for agent', action='store_true')
    add('--cycle', default=0, type=int, help='Number of cycle')
    add('--endless', default=False, help='Endless plays', action='store_true')
    add('--gamma', default=0.9, type=float, help='Discount factor')
    add('--gui', default=False, help='A simple GUI', action='store_true')
    add('--interactive', default=False, help='Text interactive interface', action='store_true')
    add('--mcts_const', default=5.0, type=float, help='PUCT constant')
    add('--mcts_sims', default=50, type=int, help='Number of MCTS sims')
    add('--mcts_tau', default=1.0, type=float, help='Temperature constant')
    add('--min_visit', default=40, type=int, help='Minimum visits for node storage')
    add('--ngames', default=50, type=int, help='Number of episodes to play')
    add('--online', default=False, help='Online agent training', action='store_true')
    add('--printboard', default=False, help='Print board', action='store_true')
    add('--print_board_to_file', default=False, help='Print board to file', action='store_true')
    add('--realtime_status', default=False, help='Save realtime game status through numpy memmap', action='store_true')
    add('--save', default=False, help='Save self-play episodes', action='store_true')
    add('--save_dir', default='./data/', type=str, help='Directory for save')
    add('--save_file', default='data', type=str, help='Filename to save')
    add('--save_tree', default=False, help='Save expanded tree nodes', action='store_true')
    add('--tetris_randomizer', default=0, type=int, help='Queue randomizer used by Tetris (0: bag, 1: uniform)')
    add('--tetris_scoring', default=0, type=int, help='Scoring system used by Tetris (0: official guideline, 1: line clears)')
    # extensions (not in the reference)
    add('--n_games', default=1, type=int, help='[new] concurrent games on the GPU')
    add('--seed', default=0, type=int, help='[new] environment seed (game g uses seed+g)')
    add('--max_moves', default=0, type=int, help='[new] stop after this many moves (0 = no limit)')
    add('--train_every', default=0, type=int, help='[new] online mode: look for harvested tuples every k moves (0: every move '
        'with one game - the reference trains at every collection, ValueSim.py:101-120 - every 10 moves with more)')
    add('--train_min_tuples', default=0, type=int, help='[new] online mode: fit only once this many fresh tuples are held over '
        'all ranks (0: 1 with one game, 1024 = one training batch with more)')
    return p


class Scoreboard:
    """Running min/max/mean/std of finished episodes, printed like the reference's ScoreTracker (play.py:25-37)."""

    def __init__(self):
        self.scores, self.lines = [], []

    def add(self, score, line):
        self.scores.append(score)
        self.lines.append(line)

    def show(self):
        s, l = np.asarray(self.scores, float), np.asarray(self.lines, float)
        print('\rGames played:{:>3}    min/max/mean/std:{:5.2f}({:5.2f})/{:5.2f}({:5.2f})/{:5.2f}({:5.2f})/{:5.2f}({:5.2f})'
              .format(len(s), s.min(), l.min(), s.max(), l.max(), s.mean(), l.mean(), s.std(), l.std()), end='', flush=True)


def main(argv=None):
    args = build_parser().parse_args(argv)
    for flag in ('save', 'save_tree', 'gui', 'interactive'):
        if getattr(args, flag):
            sys.exit('--%s is part of the reference\'s storage / UI layer (util/Data.py: PyTables / HDF5 episode files - neither '
                     'PyTables nor h5py nor libhdf5 exists in this image; util/gui.py), which this engine does not reimplement' % flag)
    if not args.agent_type:
        sys.exit('--agent_type is required (ValueSim, ValueSimLP, ValueSimC, Vanilla, VanillaC, DistValueSim)')
    from importlib import import_module
    from pyTetris import Tetris                       # the reference's own two import lines (play.py:1,81-82)
    _agent_module = import_module('agents.' + args.agent_type)

    env_args = ((20, 10), args.app, args.tetris_scoring, args.tetris_randomizer)
    G = args.n_games
    game = Tetris(*env_args, seed=args.seed, n_games=G)
    agent = getattr(_agent_module, args.agent_type)(sims=args.mcts_sims, env=Tetris, env_args=env_args, benchmark=args.benchmark,
                                             online=args.online, min_visit=args.min_visit, n_games=G)
    agent.update_root(game)

    board_file = open('board_output', 'wb') if args.print_board_to_file else None
    if args.realtime_status:
        mm = {k: np.memmap('./tmp/' + k, dtype=d, mode='w+', shape=s) for k, d, s in (
            ('board', np.int8, (20, 10)), ('combo', np.int32, (1,)), ('score', np.int32, (1,)), ('lines', np.int32, (1,)),
            ('line_stats', np.int32, (4,)))}
    board = Scoreboard()
    finished, moves = 0, 0
    while True:
        if args.printboard:
            game.printState()
        action = agent.play()
        if board_file is not None or args.realtime_status:
            st = game.getState().reshape(-1, 20, 10)[0]
            if board_file is not None:
                board_file.truncate(0)
                board_file.seek(0)
                board_file.write(st.tobytes())
                board_file.flush()
            if args.realtime_status:
                mm['board'][:] = st
                mm['combo'][:] = np.atleast_1d(game.combo)[0]
                mm['lines'][:] = np.atleast_1d(game.line_clears)[0]
                mm['score'][:] = np.atleast_1d(game.score)[0]
                mm['line_stats'][:] = np.asarray(game.line_stats).reshape(-1, 4)[0]
        game.play(action)
        agent.update_root(game)
        moves += 1
        if args.online and not args.benchmark and hasattr(agent, 'train_if_collected'):
            # the reference trains inside remove_nodes(), i.e. at every collection (ValueSim.py:101-120); the batched engine
            # harvests on the device during the move and fits here, right after a move in which a collection harvested
            # (a batch of games collects somewhere on nearly every move: it looks every few moves and waits for a batch's worth)
            agent.train_if_collected(every=args.train_every or (1 if G == 1 else 10),
                                     min_tuples=args.train_min_tuples or (1 if G == 1 else 1024))
        ended = np.atleast_1d(game.end)
        if ended.any():
            scores, lines = np.atleast_1d(game.score), np.atleast_1d(game.line_clears)
            for g in np.nonzero(ended)[0]:
                finished += 1
                if args.endless:
                    print('Episode: {:>5} Score: {:>10} Lines Cleared: {:>10}'.format(finished, int(scores[g]), int(lines[g])),
                          flush=True)
                else:
                    board.add(int(scores[g]), int(lines[g]))
            if not args.endless:
                board.show()
                if finished >= args.ngames:
                    break
            game.reset('ended')
            agent.update_root(game)
        if args.max_moves and moves >= args.max_moves:
            break
    print(flush=True)
    agent.close()


if __name__ == '__main__':
    main()
