"""Writes ksw2_comment_example.json from the alignment display quoted in the reference's comment
include/aligner/aligner_ksw2.hpp:3021-3036 (data only: three display lines per alignment)."""
import json
import os

L_T = "22333022233022233302223302223"
L_Q = "2233  222330222 3302220302220"
R_T = "33022233022233022233022233022233      0222334"
R_Q = "130222330222330220330222330222332222330222330"
G_T = "2233    3022233022233302223  30222330222330222330222330222  330222  330222330222330222330222330222330222334"
G_Q = "223322233022233022203 02220  30222330222330222330222330222  130222  330222330220330222330222332222330222330"


def cols(a, b):
    return [(x, y) for x, y in zip(a, b) if not (x == " " and y == " ")]   # a column blank in both lines is a visual separator


def cigar(c):
    out = []
    for x, y in c:
        op = "I" if x == " " else "D" if y == " " else "M"
        if out and out[-1][0] == op:
            out[-1][1] += 1
        else:
            out.append([op, 1])
    return "".join("%d%s" % (n, o) for o, n in out)


g = cols(G_T, G_Q)
doc = {
    "source": "include/aligner/aligner_ksw2.hpp:3021-3036",
    "params": {"match": 2, "mismatch": -4, "gapo": 4, "gape": 2, "end_bonus": 400},
    "left": {"target": L_T.replace(" ", ""), "query": L_Q.replace(" ", ""), "reversed_for_dp": True},
    "right": {"target": R_T.replace(" ", ""), "query": R_Q.replace(" ", "")},
    "global": {"target": "".join(x for x, y in g if x != " "), "query": "".join(y for x, y in g if y != " "),
               "cigar": cigar(g), "score": 140},
    "old_score": 130, "new_score": 140, "mem_len": 29,
}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ksw2_comment_example.json"), "w") as f:
    json.dump(doc, f, indent=1)
