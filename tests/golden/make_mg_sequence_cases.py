"""Extracts the known-answer cases of the reference's level-schedule test (tests/tp04.cc: run_tests - inputs of
get_mg_sequence / get_precondition_stmg_types and the expected sequences; tests/tp04.output records that the
reference passes every one of them) into the data fixture mg_sequence_cases.json.  Run in the build container:
    python tests/golden/make_mg_sequence_cases.py /root/reference/tests/tp04.cc
Only values are kept (no source text)."""
import json
import os
import re
import sys

src = open(sys.argv[1]).read()
body = src[src.index("run_tests()"):]
cases = []
for block in re.split(r"\n  // Test ", body)[1:]:
    def scalar(name, conv=int):
        m = re.search(name + r"\s*=\s*([^;]+);", block)
        return conv(m.group(1).strip()) if m else None

    def enum(name):
        m = re.search(name + r"\s*=\s*\w+::(\w+);", block)
        return m.group(1)

    def listof(name):
        m = re.search(name + r"\s*=\s*\{([^}]*)\}", block, re.S)
        return m.group(1) if m else None

    call = re.search(r"get_mg_sequence\((.*?)\);", block, re.S).group(1)
    args = [a.strip() for a in call.split(",")]
    # positional: n_sp_lvl, k_seq, p_seq, n_timesteps_at_once, n_min, lower_lvl, coarsening_type, time_before_space[, use_p, zip]
    trailing = [a for a in args[8:]]
    expected = [x.split("::")[1] for x in re.findall(r"MGType::\w+", listof("expected_mg_type_level"))]
    exp_p = listof("expected_p")
    cases.append({
        "name": re.search(r'"(Test[^"]*)"', block).group(1),
        "n_sp_lvl": scalar("n_sp_lvl"),
        "k_seq": [int(x) for x in listof("k_seq").split(",")],
        "n_timesteps_at_once": scalar(r"n_timesteps_at_once    "),
        "n_timesteps_at_once_min": scalar("n_timesteps_at_once_min"),
        "lower_lvl": {"tau": "t", "k": "k"}[enum("lower_lvl")],
        "coarsening_type": enum("coarsening_type"),
        "time_before_space": scalar("time_before_space", lambda s: s == "true"),
        "use_p_multigrid_space": trailing[0] == "true" if len(trailing) > 0 else False,
        "zip_from_back": trailing[1] == "true" if len(trailing) > 1 else True,
        "expected_mg_type_level": "".join({"tau": "t", "k": "k", "h": "h", "p": "p"}[x] for x in expected),
        "expected_precondition_types": [int(x) for x in exp_p.split(",")] if exp_p else None,
    })
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mg_sequence_cases.json")
json.dump(cases, open(out, "w"), indent=1)
print(len(cases), "cases ->", out)
