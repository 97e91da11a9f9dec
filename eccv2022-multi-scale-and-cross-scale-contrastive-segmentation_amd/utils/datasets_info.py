"""Dataset class tables: ``DATASETS_INFO[dataset].CLASS_INFO[experiment] = [remap, names, categories]``
with the same access pattern as the reference (utils/defaults.py:1, utils/datasets_info/*.py).
The tables themselves are data (``datasets_info.json``, dumped by tools/gen_datasets_info.py);
``names`` is an id -> class-name dict whose length is ``num_all_classes`` and whose key 255 marks
the ignore class."""
import json
import os


class _Info(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _restore(x):
    if isinstance(x, dict) and "__dict__" in x:
        out = {}
        for k, v in x["__dict__"]:
            out[tuple(k) if isinstance(k, list) else k] = _restore(v)
        return out
    if isinstance(x, list):
        return [_restore(v) for v in x]
    return x


def _load():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "datasets_info.json")
    with open(path) as f:
        raw = json.load(f)
    info = {}
    for ds, class_info in raw.items():
        ci = [[_restore(p) for p in exp] for exp in class_info]
        names = [[exp[1][k] for k in sorted(exp[1].keys())] for exp in ci]
        info[ds] = _Info(CLASS_INFO=ci, CLASS_NAMES=names)
    return info


DATASETS_INFO = _Info(_load())


def register_dataset(name, class_names, ignore=True):
    """Add a synthetic dataset entry (e.g. BASELINE config 1: 3 real classes + ignore).
    ``class_names``: list of real class names; ids 0..len-1, plus 255 -> 'Ignore' if ``ignore``."""
    names = {i: n for i, n in enumerate(class_names)}
    remap = {i: [i] for i in range(len(class_names))}
    if ignore:
        names[255] = "Ignore"
        remap[255] = [255]
    cats = {"all": list(range(len(class_names)))}
    entry = [remap, names, cats]
    DATASETS_INFO[name] = _Info(CLASS_INFO=[entry, entry],
                                CLASS_NAMES=[[names[k] for k in sorted(names)]] * 2)
    return DATASETS_INFO[name]


def num_all_classes(dataset, experiment):
    return len(DATASETS_INFO[dataset].CLASS_INFO[experiment][1])


def ignore_class(dataset, experiment):
    names = DATASETS_INFO[dataset].CLASS_INFO[experiment][1]
    return (len(names) - 1) if 255 in names else -1
