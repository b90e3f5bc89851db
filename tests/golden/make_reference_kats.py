"""Writes tests/golden/reference_kats.json.

The vectors are DATA transcribed by hand from the reference's own test files (inputs and expected
outputs only -- no reference source text); each entry cites the file:line it comes from.  The
reference is Julia and cannot be run in this image (no julia binary), so nothing here was produced
by executing the reference.  Entries under "derived" are NOT from the reference: they are
hand-derivable known answers for operators whose prox the reference's tests leave unpinned
(SURVEY.md Appendix B), kept separate so the provenance stays visible.
"""
import json
import os

NU = 1 / 9.1e4
QRAW = [2631.441298528196, -533.9101219466443, 466.56156501426733, 1770.8953574224836, -2554.7769423950244]
Q5 = [-NU * v for v in QRAW]

kats = {
    "provenance": "transcribed from /root/reference/test (v0.2.2); see cite fields",
    "box_golden": {
        "cite": "test/runtests.jl:417-494 (n=5, h=Op(1.0), x=ones(5), Delta=0.01, nu=1/9.1e4, once shifted)",
        "n": 5, "lambda": 1.0, "x": [1.0] * 5, "s": [0.0] * 5, "delta": 0.01, "sigma": NU, "q": Q5,
        "rtol": 1.4901161193847656e-08,
        "expected": {
            "ShiftedNormL0Box": [-0.010000000000000, 0.005867144197216, -0.005127050164992, -0.010000000000000, 0.010000000000000],
            "ShiftedNormL1Box": [-0.010000000000000, 0.005856155186227, -0.005138039175981, -0.010000000000000, 0.010000000000000],
            "ShiftedRootNormLhalfBox": [-0.010000000000000, 0.005861665724748, -0.005132558825434, -0.010000000000000, 0.010000000000000],
            "ShiftedNormL1B2": [-0.006367076930786, 0.001288947922799, -0.001130889587543, -0.004285677352167, 0.006176811716709],
        },
    },
    "group_l2_binf_single": {
        "cite": "test/runtests.jl:555-606 (NormL2(1.0) -> one group [:], x=ones(5), Delta=0.01)",
        "n": 5, "lambda": [1.0], "offsets": [0, 5], "x": [1.0] * 5, "s": [0.0] * 5, "delta": 0.01,
        "sigma": NU, "q": Q5, "rtol": 1.4901161193847656e-08,
        "expected": [-0.010000000000000, 0.005862191941930, -0.005131948291800, -0.010000000000000, 0.010000000000000],
    },
    "group_l2_binf_two": {
        "cite": "test/runtests.jl:655-705 (v=[1:3,4:6])",
        "n": 6, "lambda": [0.396767474230670, 0.538816734003357], "offsets": [0, 3, 6], "x": [1.0] * 6,
        "s": [0.0] * 6, "delta": 0.01, "sigma": 0.419194514403295,
        "q": [-0.649013765191241, 1.181166041965532, -0.758453297283692, -1.109613038501522, -0.845551240007797, -0.572664866457950],
        "rtol": 1.4901161193847656e-08,
        "expected": [-0.010000000000000, 0.010000000000000, -0.010000000000000, -0.010000000000000, -0.010000000000000, -0.010000000000000],
    },
    "rootnormlhalf_unshifted": {
        "cite": "test/runtests.jl:113-126 (sum of squared errors <= 1e-11)",
        "q": [0.1097, 1.1287, -0.29, 1.2616], "lambda": 0.7788, "nu": 0.1056,
        "expected": [0.0, 1.0893, -0.197463, 1.22444], "sumsq_tol": 1e-11,
    },
    "testsbox": {
        "cite": "test/testsbox.jl:1-99 (scalar cases; psi=shifted(h,x,l,u); omega=shifted(psi,s); atol=1e-2)",
        "sigma": 1.0, "l": 0.0, "u": 3.0, "s": -1.0, "atol": 1.0e-2,
        "ShiftedNormL0Box": {
            "q": [5.0, 5.0, 5.0, 0.0, 0.0, 0.0, 3.0, 3.0, 3.0],
            "x": [1.0, -1.0, -1.0, 1.0, -1.0, -1.0, 1.0, -1.0, -1.0],
            "lambda": [1.0, 5.0, 3.0, 1.0, 2.0, 1.0, 1.0, 1.0, 0.1],
            "sol": [4.0, 2.0, 4.0, 1.0, 2.0, 1.0, 3.0, 2.0, 3.0],
        },
        "ShiftedNormL1Box": {
            "q": [0.5, 5.0, 3.0, -2.0, 4.0, 1.0, 1.0, 7.0, 4.0],
            "x": [1.0, -4.0, -2.0, -1.0, -5.0, -3.0, 3.0, -2.0, 1.0],
            "lambda": [1.0] * 9,
            "sol": [1.0, 4.0, 3.0, 1.0, 4.0, 2.0, 1.0, 4.0, 3.0],
        },
        "ShiftedRootNormLhalfBox": {
            "q": [5.0, 5.0, 5.0, 2.0, 0.0, 1.0, 0.0, 3.0, 3.0],
            "x": [1.0, -1.0, -1.0, 1.0, 1.0, -1.0, -1.0, -1.0, -1.0],
            "lambda": [1.0, 10.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.5, 1.0],
            "sol": [4.0, 2.0, 4.0, 1.6054, 1.0, 2.0, 1.0, 2.702, 2.0],
        },
    },
    "group_l2_vs_norml2": {
        "cite": "test/runtests.jl:213-251,286-329 (property: ShiftedGroupNormL2 prox == per-group NormL2 prox of q+x, minus x; 2-norm <= 1e-11)",
        "tol": 1e-11,
    },
    "iprox_testsbox": {
        "cite": "test/testsbox.jl:101-304 (14 scalar cases per operator; psi=shifted(h,x,l,u); omega=shifted(psi,s); iprox(omega,g,d); exact ==)",
        "l": -2.0, "u": 1.0, "s": -1.0,
        "ShiftedNormL0Box": {
            "d": [0.0, 0.0, 0.0, 0.0, 0.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, -2.0, -2.0, -2.0],
            "g": [0.0, 0.0, 2.0, 2.0, -2.0, 1.0, 0.0, 1.0, 10.0, -10.0, 4.0, -10.0, 10.0, -4.0],
            "x": [0.0, -10.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
            "lambda": [1.0, 1.0, 1.0, 10.0, 1.0, 1.0, 0.1, 10.0, 1.0, 1.0, 10.0, 1.0, 1.0, 10.0],
            "sol": [1.0, 0.0, -1.0, 1.0, 2.0, -0.5, 0.0, 1.0, -1.0, 2.0, 1.0, 2.0, -1.0, 1.0],
        },
        "ShiftedNormL1Box": {
            "d": [0.0, 0.0, 0.0, 0.0, 0.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, -2.0, -2.0, -2.0],
            "g": [0.5, 0.5, 0.5, 2.0, -2.0, 0.0, 1.0, 1.0, -1.0, 1.0, 1.0, 0.0, 1.0, 1.0],
            "x": [0.0, 4.0, -2.0, 0.0, 0.0, 4.0, -2.0, 1.0, 0.5, 0.5, 3.0, 1.0, 1.0, 1.0],
            "lambda": [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 10.0, 1.0],
            "sol": [1.0, -1.0, 2.0, -1.0, 2.0, -0.5, 0.0, 0.0, 0.5, 0.0, -1.0, 2.0, 0.0, -1.0],
        },
    },
    "derived": {
        "provenance": "NOT from the reference: hand-derivable answers (SURVEY.md Appendix B) for operators whose prox values the reference tests leave as TODO (runtests.jl:179-180,382-383,772-773)",
        "setA_unboxed": {
            "x": [1.0] * 5, "s": [0.0] * 5, "lambda": 1.0, "sigma": NU, "q": Q5,
            "ShiftedNormL1": [-0.028927926357452702, 0.0058561551862268595, -0.0051380391759809595, -0.019471377554093224, 0.028063482883461804],
            "ShiftedNormL0": [-0.02891693734646369, 0.005867144197215871, -0.005127050164991948, -0.019460388543104213, 0.028074471894450816],
            "ShiftedRootNormLhalf": [-0.028922513075614886, 0.005861665724747667, -0.005132558825434397, -0.019465937320080173, 0.02806905291543904],
        },
        "setB": {
            "x": [0.5, -1.25, 2.0, -0.125, 0.75, -3.0, 0.0, 1.5],
            "s": [0.25, 0.5, -0.5, 0.125, -0.25, 0.0, 0.0, -1.0],
            "q": [-1.0, 0.5, 0.25, 3.0, -0.5, 2.0, 0.1, -0.75],
            "lambda": 0.5, "sigma": 0.8,
            "ShiftedNormL1": [-0.75, 0.75, -0.15000000000000002, 2.6, -0.5, 2.4, -0.0, -0.5],
            "ShiftedNormL0": [-0.75, 0.75, 0.25, 3.0, -0.5, 2.0, -0.0, -0.5],
            "ShiftedRootNormLhalf": [-0.75, 0.75, 0.09146258246265293, 2.88219372844361, -0.5, 2.227561040322781, 0.0, -0.5],
            "ShiftedIndBallL0_r3": [-0.75, 0.75, 0.25, 3.0, -0.5, 2.0, 0.0, -0.5],
            "ShiftedIndBallL0BInf_r3_delta0.6": [-0.6, 0.6, 0.25, 0.6, -0.5, 0.6, 0.0, -0.5],
        },
        "tiebreak": {
            "x": [0.0] * 6, "s": [0.0] * 6, "q": [2.0, -2.0, 1.0, 2.0, -1.0, 0.5],
            "r2": [2.0, -2.0, 0.0, 0.0, 0.0, 0.0],
            "r3": [2.0, -2.0, 0.0, 2.0, 0.0, 0.0],
            "r4": [2.0, -2.0, 1.0, 2.0, 0.0, 0.0],
        },
    },
}

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
    with open(out, "w") as f:
        json.dump(kats, f, indent=1)
    print("wrote", out)
