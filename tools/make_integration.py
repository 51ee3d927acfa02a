"""Regenerate section 1 of INTEGRATION.md from oracle/dropin/kalign_amd_glue.c (the compiled glue), so that the
replacement bodies quoted there are the text that is built and tested.  Usage: python tools/make_integration.py"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "oracle", "dropin", "kalign_amd_glue.c")
DOC = os.path.join(ROOT, "INTEGRATION.md")

# (heading, functions quoted under it: the comment block in front of the first one is included)
SEAMS = [
    ("1a. `create_msa_tree` (`lib/src/aln_run.c:43-78`) and `create_msa_tree_inline_refine` (`lib/src/aln_run.c:448-475`)",
     ["glue_tree", "glue_collect", "create_msa_tree", "create_msa_tree_inline_refine"]),
    ("1b. `refine_alignment` (`lib/src/aln_refine.c:36-88`)", ["refine_alignment"]),
    ("1c. `anchor_consistency_build` (`lib/src/anchor_consistency.c:200-275`)", ["anchor_consistency_build"]),
    ("1d. `build_tree_kmeans` (`lib/src/bisectingKmeans.c:177-271`) and `build_tree_kmeans_noisy` (`:76-175`)",
     ["glue_kmeans", "build_tree_kmeans", "build_tree_kmeans_noisy"]),
    ("1e. `finalise_alignment` (`lib/src/msa_op.c:546-576`)", ["finalise_alignment"]),
    ("1f. `compute_aln_pairwise_dist` (`lib/src/aln_apair_dist.c:9-86`) and `build_tree_from_pairwise` (`lib/src/bisectingKmeans.c:1150-1200`)",
     ["compute_aln_pairwise_dist", "build_tree_from_pairwise"]),
    ("1g. The member loop of `kalign_ensemble` (`lib/src/ensemble.c:286-339`): members side by side, one device each",
     ["glue_member_thread", "glue_member_take", "kalign_amd_member_run_seeded", "kalign_amd_member_run_realign", "kalign_ensemble"]),
]


def function_text(src, name):
    """the definition of `name` (not a prototype) with the comment block directly above it"""
    m = None
    for m in re.finditer(r"^(?:static )?(?:int|void\*?) %s\([^;{]*\)\n\{" % re.escape(name), src, re.M):
        break
    assert m, name
    start = m.start()
    head = src[:start].rstrip("\n")
    if head.endswith("*/"):
        start = head.rindex("/*")
    depth, i = 0, m.end() - 1
    while True:
        if src[i] == "{":
            depth += 1
        elif src[i] == "}":
            depth -= 1
            if depth == 0:
                break
        i += 1
    return src[start:i + 1]


def render():
    """INTEGRATION.md with section 1 regenerated from the glue file"""
    src = open(GLUE).read()
    doc = open(DOC).read()
    a = doc.index("### 1a.")
    b = doc.index("Notes\n", a)
    parts = []
    for heading, fns in SEAMS:
        parts.append("### %s\n\n```c\n%s\n```\n" % (heading, "\n\n".join(function_text(src, f) for f in fns)))
    return doc[:a] + "\n".join(parts) + "\n" + doc[b:]


def main():
    import sys
    doc = render()
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        if doc != open(DOC).read():
            sys.exit("INTEGRATION.md section 1 is not the text of %s: run python tools/make_integration.py" % os.path.relpath(GLUE, ROOT))
        print("INTEGRATION.md section 1 == the compiled glue")
        return
    open(DOC, "w").write(doc)
    print("INTEGRATION.md: section 1 regenerated from", os.path.relpath(GLUE, ROOT))


if __name__ == "__main__":
    main()
