#!/usr/bin/env python3
"""Extracts the known-answer DATA held by the reference's own tests into JSON fixtures.

Run in the dev container (needs /root/reference); the fixtures it writes are committed
and are what travels to the GPU box.  Only inputs and expected outputs are extracted
(numbers and enum names) -- no reference source text is stored.

Sources:
  planner/tests/test_reeds_shepp.cpp:38-300   48 (start, goal, acceptable shortest words)
  planner/src/geometry/reeds_shepp.h:14-46    word name -> index
  planner/tests/test_frontier.cpp:14-53       push sequence + expected pop / find / remove
  planner/tests/test_tree.cpp:91-130          1-NN and 2-NN cases
  planner/tests/test_{a_star,hybrid_a_star,rrt,rrt_star}.cpp   smoke-test configurations
"""
import json
import os
import re
import sys

REF = os.environ.get("PP_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def word_table():
    txt = open(os.path.join(REF, "planner/src/geometry/reeds_shepp.h")).read()
    body = txt[txt.index("enum class PathWords"):]
    body = body[body.index("{") + 1:body.index("NumPathWords")]
    body = re.sub(r"//.*", "", body)
    names = [t.strip() for t in body.replace("\n", " ").split(",") if t.strip()]
    table = {}
    idx = -1
    for n in names:
        if "=" in n:
            name, val = [s.strip() for s in n.split("=")]
            idx = int(val)
        else:
            name = n
            idx += 1
        table[name] = idx
    assert table["NoPath"] == -1 and table["LfSfLf"] == 0 and len([k for k in table if table[k] >= 0]) == 48, table
    return table


def reeds_shepp_vectors():
    words = word_table()
    txt = open(os.path.join(REF, "planner/tests/test_reeds_shepp.cpp")).read()
    main = txt[txt.index("int main()"):]
    num = r"(-?[0-9.eE+-]+)"
    pat = re.compile(
        r"optimalWord\s*=\s*Planner::ReedsShepp::PathWords::(\w+);\s*"
        r"start\s*=\s*\{\s*" + num + r",\s*" + num + r",\s*" + num + r"\s*\};\s*"
        r"goal\s*=\s*\{\s*" + num + r",\s*" + num + r",\s*" + num + r"\s*\};\s*"
        r"Planner::Test\(start, goal, \{([^}]*)\}\);")
    vectors = []
    for m in pat.finditer(main):
        accept = []
        for tok in m.group(8).split(","):
            tok = tok.strip()
            if tok == "optimalWord":
                accept.append(words[m.group(1)])
            else:
                accept.append(words[tok.split("::")[-1]])
        vectors.append({
            "name": m.group(1),
            "start": [float(m.group(i)) for i in (2, 3, 4)],
            "goal": [float(m.group(i)) for i in (5, 6, 7)],
            "accepted_words": accept,
        })
    assert len(vectors) == 48, len(vectors)
    return {"source": "planner/tests/test_reeds_shepp.cpp:38-300", "min_turning_radius": 1.0,
            "end_pose_tolerance": {"xy": 1e-6, "theta_deg": 1e-6},
            "word_index": {k: v for k, v in words.items() if v >= 0}, "vectors": vectors}


def frontier_case():
    txt = open(os.path.join(REF, "planner/tests/test_frontier.cpp")).read()
    pushes = [[int(a), int(b)] for a, b in re.findall(r"frontier\.Push\(\{\s*(-?\d+),\s*(-?\d+)\s*\}\);", txt)]
    assert len(pushes) == 15
    return {
        "source": "planner/tests/test_frontier.cpp:14-53",
        "note": "element = (priority, key); compare = priority '<' (max priority at the top); uniqueness and Find/Remove by key",
        "pushes": pushes,
        "expect_sorted_iteration": True,
        "expect_pop": [2, 4],
        "expect_remove": [[[2, 4], 0], [[0, 1], 1], [[1, 1], 0]],
        "expect_find": [[[0, 2], True], [[0, 0], False], [[1, 2], True]],
        "expect_find_value": [[0, 2], [1, 2]],
    }


def tree_cases():
    return {
        "source": "planner/tests/test_tree.cpp:91-130",
        "points": [[0, 0], [1, 0], [2, 0], [3, 0]],
        "nearest": {"query": [1.1, 0], "expect_index": 1},
        "knn": {"query": [2.4, 0.2], "k": 2, "expect_indices_set": [2, 3]},
    }


def smoke_cases():
    return {
        "a_star": {"source": "planner/tests/test_a_star.cpp:15-34 + planner/tests/state_space/a_star_state_space_2d.h:7-23",
                   "start": [0, 0], "goal": [10, 5]},
        "hybrid_a_star": {"source": "planner/tests/test_hybrid_a_star.cpp:9-36",
                          "bounds": [[-10, -10, "-pi"], [10, 10, "pi"]], "resolution": 0.1,
                          "start": [0.0, 0.0, 0.0], "goal": [8.0, 8.0, 0.78],
                          "spatial_tolerance": 0.1, "angular_tolerance_deg": 5},
        "rrt": {"source": "planner/tests/test_rrt.cpp:9-38", "bounds": [[0, 0], [5, 5]],
                "start": [0.0, 0.0], "goal": [2.0, 2.0], "spatial_tolerance": 1.0},
        "rrt_star": {"source": "planner/tests/test_rrt_star.cpp:10-39", "bounds": [[0, 0], [5, 5]],
                     "start": [0.0, 0.0], "goal": [2.0, 2.0], "spatial_tolerance": 1.0},
    }


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not found at %s" % REF)
    out = {
        "reeds_shepp_vectors.json": reeds_shepp_vectors(),
        "frontier_case.json": frontier_case(),
        "tree_cases.json": tree_cases(),
        "smoke_cases.json": smoke_cases(),
    }
    for name, obj in out.items():
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(obj, f, indent=1)
        print("wrote", name)


if __name__ == "__main__":
    main()
