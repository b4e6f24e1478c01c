"""Extracts the reference's ONLY recorded ground truth on this path into a data fixture.

Run in the build container (needs /root/reference; nothing here travels to the GPU box except the JSON):

    python tests/golden/make_tf_variables.py

Source: the saved cell outputs of /root/reference/notebooks/play.ipynb — (i) the flag listing and the
`print_variables_by_scope()` / `print_num_params()` output of the shipped MLP-SQAIR config (lines 239-362 of the
notebook file: 105 trainable variables, 2 951 522 parameters), (ii) the validation metrics of the released
1M-iteration checkpoint (line ~480).  The output `tf_variables.json` holds names, shapes, per-scope totals, the
flags and that one metric record — data, not source.
"""
import json
import os
import re
import sys

REF = os.environ.get("SQAIR_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def cell_text(cell):
    out = []
    for o in cell.get("outputs", []):
        if "text" in o:
            out.append("".join(o["text"]))
    return "".join(out)


def main():
    nb = json.load(open(os.path.join(REF, "notebooks", "play.ipynb")))
    listing = metrics = None
    for c in nb["cells"]:
        if c["cell_type"] != "code":
            continue
        t = cell_text(c)
        if "Trainable Variables:" in t:
            listing = t
        if "Data validation" in t:
            metrics = t
    assert listing and metrics
    flags, variables, scopes, order = {}, [], {}, []
    in_flags = False
    scope = None
    for line in listing.splitlines():
        s = line.strip()
        if s == "Flags:":
            in_flags = True
            continue
        if s.startswith("Trainable Variables"):
            in_flags = False
            continue
        if in_flags:
            m = re.match(r"(\w+): (.*)$", s)
            if m:
                flags[m.group(1)] = m.group(2)
            continue
        m = re.match(r"scope: (\w+)$", s)
        if m:
            scope = m.group(1)
            continue
        m = re.match(r"(\S+):0 \[(.*)\]$", s)
        if m:
            shape = [int(v) for v in m.group(2).split(",")] if m.group(2).strip() else []
            variables.append(dict(name=m.group(1), shape=shape, scope=scope))
            continue
        m = re.match(r"(\w+) scope params = ([\d ]+)$", s)
        if m:
            scopes[m.group(1)] = int(m.group(2).replace(" ", ""))
            continue
        m = re.match(r"Number of trainable parameters: ([\d ]+)$", s)
        if m:
            total = int(m.group(1).replace(" ", ""))
    rec = {}
    m = re.search(r"Step (\d+), Data validation (.*), eval time", metrics)
    rec["step"] = int(m.group(1))
    for kv in m.group(2).split(", "):
        k, v = kv.split(" = ")
        rec[k] = float(v)
    out = dict(
        source="notebooks/play.ipynb (cell outputs): print_flags / print_variables_by_scope / print_num_params and the "
               "validation record of release_models/mnist_mlp/1/model.ckpt-1000000",
        flags=flags, variables=variables, scope_totals=scopes, total=total, validation_record=rec,
        # how the record's numbers are normalised (model.py:88-135, eval_tools.make_expr_logger): elbo_* are sums over
        # the T = 10 frames of a sequence, averaged over sequences; data_ll / kl / log_p_z / log_q_z_given_x / num_*steps
        # are importance-weighted means over particles of per-FRAME means (model.py:202-205, `_imp_weighted_mean` of
        # reduce_mean over time), hence data_ll - kl = 609.66 ~ elbo_iwae / 10 = 609.55
        record_normalisation=dict(per_sequence=["elbo_vae", "elbo_iwae"],
                                  per_frame=["data_ll", "kl", "log_p_z", "log_q_z_given_x", "num_steps/t",
                                             "num_disc_steps/t", "num_prop_steps/t"],
                                  fraction=["num_steps_acc"], seq_len=10),
    )
    assert len(variables) == 105 and sum(
        (1 if not v["shape"] else __import__("functools").reduce(lambda a, b: a * b, v["shape"])) for v in variables) == total
    path = os.path.join(HERE, "tf_variables.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(path, len(variables), total, rec)


if __name__ == "__main__":
    sys.exit(main())
