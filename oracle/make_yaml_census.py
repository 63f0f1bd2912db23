"""TEST INFRASTRUCTURE — census of the reference's training YAMLs (configs/*.yaml with an ``encoder_configs`` key).

Run only in the build container (needs /root/reference):
    python oracle/make_yaml_census.py

Writes tests/golden/ref_yaml_census.json: per file the list of top-level keys, the scalar / list settings the model and the
training loop read (utils/config.py:9-61,96-117), and per modality the encoder type, sizes and the collator settings
(type, pad_len, dropout ...).  No YAML text is stored: dataset paths, output directories and wandb names are dropped.
tests/test_aux_cpu.py rebuilds a config from each entry and checks that the native config loader / model constructor
accept it (every training YAML of the reference runs unmodified).
"""
import glob
import json
import os

import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROP = {"dataset", "output_dir", "restart", "wandb_name", "wandb_account_name", "wandb_restart"}

out = {}
for f in sorted(glob.glob("/root/reference/configs/*.yaml")):
    d = yaml.safe_load(open(f)) or {}
    if "encoder_configs" not in d:
        continue
    ent = {"keys": sorted(d.keys()),
           "settings": {k: v for k, v in d.items() if k not in DROP and not isinstance(v, dict)},
           "has_restart": bool(d.get("restart")),
           "encoder_configs": d["encoder_configs"],
           "modality_config": d.get("modality_config", {})}
    out[os.path.basename(f)] = ent
path = os.path.join(REPO, "tests", "golden", "ref_yaml_census.json")
json.dump(out, open(path, "w"), indent=0, sort_keys=True)
print(len(out), "training configs ->", path, os.path.getsize(path), "bytes")
