"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the reference's front end: one circuit description run twice.

Restates, in plain Python,

  * plonk::builder::{Context, BuildVar}      /root/reference/plonk/src/builder.rs:128-195, 326-369 (ids, tags, the
                                             deferred equalities, `finish`)
  * plonk::builder::ComputeVar               /root/reference/plonk/src/builder.rs:371-396, 435-440 (witness recording;
                                             assert_eq is a no-op)
  * CircuitBuilder::fill                     /root/reference/plonk/src/builder.rs:47-58 (first power of two >= gates + 3, from 2)
  * Gate::to_row                             /root/reference/plonk/src/builder.rs:314-324
  * PermutationBuilder::{add_constrain, build}  /root/reference/permutation/src/lib.rs:46-93

so that the C++ front end (tests/cpp/circuit_host.hpp) can be compared with it on arbitrary circuits.  The
reference iterates its constraints in HashMap order (lib.rs:68): the partition into cycles is defined, the order inside
a cycle is not -- comparisons are made on the partition.  Pinned by the README circuit, whose tables SURVEY.md KAT-5
lists (tests/test_oracle.py).  Only tests/ may import it.
"""
from __future__ import annotations

from . import bls12_381 as O

R = O.R
MUL, ADD, DUMMY = "M", "A", "D"
ROWS = {MUL: [0, 0, 1, 1, 0], ADD: [1, 1, 1, 0, 0], DUMMY: [0, 0, 0, 0, 0]}   # q_l q_r q_o q_m q_c


class BuildContext:
    def __init__(self):
        self.gates, self.constraints, self.pending = [], [], []
        self.next_id, self.tag = 0, {}

    def new_id(self):
        self.next_id += 1
        return self.next_id - 1

    def add_eq(self, left, right):                       # builder.rs:152-169
        a, b = self.tag.get(left), self.tag.get(right)
        if a is not None and b is not None:
            self.constraints.append((a, b))
        else:
            self.pending.append((left, right))

    def finish(self):                                    # builder.rs:170-195
        waiting, self.pending = self.pending, []
        for l, r in waiting:
            self.add_eq(l, r)
        if self.pending:
            raise ValueError("assert_eq on a variable that never enters a gate")
        size = 2
        while size < len(self.gates) + 3:
            size *= 2
        self.gates += [DUMMY] * (size - len(self.gates))
        return self.gates, self.constraints


class BuildVar:
    def __init__(self, cx, vid):
        self.cx, self.id = cx, vid

    def _gate(self, rhs, kind):                          # builder.rs:339-369
        cx = self.cx
        cx.gates.append(kind)
        j = len(cx.gates) - 1
        out = cx.new_id()
        cx.tag[out] = (2, j)
        for col, vid in ((0, self.id), (1, rhs.id)):
            if vid in cx.tag:
                copy = cx.new_id()
                cx.tag[copy] = (col, j)
                cx.add_eq(vid, copy)
            else:
                cx.tag[vid] = (col, j)
        return BuildVar(cx, out)

    def __add__(self, o):
        return self._gate(o, ADD)

    def __mul__(self, o):
        return self._gate(o, MUL)

    def assert_eq(self, o):
        self.cx.add_eq(self.id, o.id)


class ComputeVar:
    def __init__(self, value, advice):
        self.value, self.advice = value % R, advice

    def _rec(self, o, out):
        for col, v in zip(self.advice, (self.value, o.value, out)):
            col.append(v)
        return ComputeVar(out, self.advice)

    def __add__(self, o):
        return self._rec(o, (self.value + o.value) % R)

    def __mul__(self, o):
        return self._rec(o, self.value * o.value % R)

    def assert_eq(self, o):
        pass


def build_permutation(constraints, size):
    """PermutationBuilder::build (lib.rs:62-93): merge the cycles of every constrained pair by exchanging successors,
    the smaller cycle relabelled.  Returns the successor map over flat indices j + i * size."""
    n = 3 * size
    mapping, aux, sizes = list(range(n)), list(range(n)), [1] * n
    for (li, lj), (ri, rj) in constraints:
        left, right = lj + li * size, rj + ri * size
        if aux[left] == aux[right]:
            continue
        if sizes[aux[left]] < sizes[aux[right]]:
            left, right = right, left
        sizes[aux[left]] += sizes[aux[right]]
        nxt, label = right, aux[left]
        while True:
            aux[nxt] = label
            nxt = mapping[nxt]
            if aux[nxt] == label:
                break
        mapping[left], mapping[right] = mapping[right], mapping[left]
    return mapping


def compile_circuit(run, n_inputs):
    """CircuitBuilder::compile up to the tables: (rows, gate kinds, selector rows, permutation)"""
    cx = BuildContext()
    run([BuildVar(cx, cx.new_id()) for _ in range(n_inputs)])
    gates, constraints = cx.finish()
    perm = build_permutation(constraints, len(gates))
    return len(gates), gates, [ROWS[g] for g in gates], perm


def witness(run, inputs):
    """CompiledCircuit::prove up to the unpadded, unblinded witness columns (proof.rs:33-41)"""
    advice = ([], [], [])
    run([ComputeVar(v, advice) for v in inputs])
    return advice


def cycles(perm):
    seen, out = set(), set()
    for s in range(len(perm)):
        if s in seen:
            continue
        cyc, k = [], s
        while k not in seen:
            seen.add(k)
            cyc.append(k)
            k = perm[k]
        out.add(frozenset(cyc))
    return out


def random_program(seed, n_inputs, n_ops):
    """A circuit description as data -- the same generator exists in tests/cpp/test_circuit_tables_host.cpp:
    x <- x * 6364136223846793005 + 1442695040888963407 (mod 2^64), take x >> 33.  Each step: kind = r % 8
    (0..2 add, 3..5 mul, 6..7 assert_eq), operands = two draws modulo the number of variables so far."""
    x, prog, nvars = seed & (2**64 - 1), [], n_inputs

    def draw():
        nonlocal x
        x = (x * 6364136223846793005 + 1442695040888963407) & (2**64 - 1)
        return x >> 33

    for _ in range(n_ops):
        kind = draw() % 8
        a, b = draw() % nvars, draw() % nvars
        if kind <= 5:
            prog.append(("add" if kind <= 2 else "mul", a, b))
            nvars += 1
        else:
            prog.append(("eq", a, b))
    return prog


def run_program(prog):
    def run(inputs):
        v = list(inputs)
        for op, a, b in prog:
            if op == "add":
                v.append(v[a] + v[b])
            elif op == "mul":
                v.append(v[a] * v[b])
            else:
                v[a].assert_eq(v[b])
    return run
