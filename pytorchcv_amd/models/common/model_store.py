"""
    Weight store with the reference's contract (pytorchcv/models/common/model_store.py:140-387): files named
    `{model}-{top5err:04d}-{sha1[:8]}.pth` under `~/.torch/models`, SHA-1 checked, fetched as `.pth.zip` from the imgclsmob
    GitHub releases when absent, loaded with `torch.load` and filtered to the keys the net owns. The index
    (`model_metainfos.csv`) carries the rows of the model families this package builds. The build pipeline has no network, so
    the download branch is exercised only by users.
"""

__all__ = ['get_model_metainfo_dict', 'get_model_file', 'load_model', 'download_model', 'calc_net_weight_count',
           'get_model_weight_count']

import os
import csv
import zipfile
import logging
import hashlib

imgclsmob_repo_url = "https://github.com/osmr/imgclsmob"
_INDEX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_metainfos.csv")


def get_model_metainfo_dict():
    """name -> (trainable weight count, 4-digit error string, sha1, release tag)."""
    with open(_INDEX, "r", newline="") as f:
        rows = list(csv.reader(f))
    return {r[0]: (int(r[1]) if r[1] != "NA" else 0, r[2], r[3], r[4]) for r in rows[1:]}


def _metainfo(model_name):
    table = get_model_metainfo_dict()
    if model_name not in table:
        raise ValueError("Pretrained model for {name} is not available.".format(name=model_name))
    return table[model_name]


def get_model_weight_count(model_name):
    return _metainfo(model_name)[0]


def _sha1_ok(path, sha1_hash):
    h = hashlib.sha1()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest() == sha1_hash


def _fetch(url, path, retries=5):
    import requests
    last = None
    for attempt in range(retries + 1):
        try:
            r = requests.get(url, stream=True, timeout=60)
            if r.status_code != 200:
                raise RuntimeError("Failed downloading url {}".format(url))
            with open(path, "wb") as f:
                for chunk in r.iter_content(chunk_size=1 << 16):
                    if chunk:
                        f.write(chunk)
            return path
        except Exception as e:      # noqa: BLE001 - retried, re-raised below
            last = e
            print("download failed, retrying, {} attempt{} left".format(retries - attempt, "s" if retries - attempt != 1 else ""))
    raise last


def get_model_file(model_name, local_model_store_dir_path=os.path.join("~", ".torch", "models")):
    """Path of the verified local `.pth`, downloading it first if it is missing or corrupt."""
    _, error, sha1_hash, tag = _metainfo(model_name)
    file_name = "{}-{}-{}.pth".format(model_name, error, sha1_hash[:8])
    root = os.path.expanduser(local_model_store_dir_path)
    file_path = os.path.join(root, file_name)
    if os.path.exists(file_path):
        if _sha1_ok(file_path, sha1_hash):
            return file_path
        logging.warning("Mismatch in the content of model file detected. Downloading again.")
    else:
        logging.info("Model file not found. Downloading to {}.".format(file_path))
    os.makedirs(root, exist_ok=True)
    zip_path = file_path + ".zip"
    _fetch("{}/releases/download/{}/{}.zip".format(imgclsmob_repo_url, tag, file_name), zip_path)
    # only the one expected member leaves the archive, under its expected name (no path-traversal names, no other files),
    # into a temporary file that becomes the model file only after its SHA-1 has been verified
    tmp_path = file_path + ".part"
    try:
        with zipfile.ZipFile(zip_path) as zf:
            if file_name not in zf.namelist():
                raise ValueError("Downloaded archive does not contain {}".format(file_name))
            with zf.open(file_name) as src, open(tmp_path, "wb") as dst:
                for chunk in iter(lambda: src.read(1 << 20), b""):
                    dst.write(chunk)
    finally:
        os.remove(zip_path)
    if _sha1_ok(tmp_path, sha1_hash):
        os.replace(tmp_path, file_path)
        return file_path
    os.remove(tmp_path)
    raise ValueError("Downloaded file has different hash. Please try again.")


def load_model(net, file_path, ignore_extra=True):
    """Load a `.pth` state dict; with `ignore_extra`, entries the net does not own are dropped first."""
    import torch
    # the released checkpoints are plain tensor state dicts: the restricted unpickler is all they need (the reference passes
    # weights_only=False, model_store.py:325, which would run arbitrary pickle code from a downloaded file)
    state = torch.load(file_path, weights_only=True, map_location="cpu")
    if ignore_extra:
        own = net.state_dict()
        state = {k: v for k, v in state.items() if k in own}
    net.load_state_dict(state)


def download_model(net, model_name, local_model_store_dir_path=os.path.join("~", ".torch", "models"), ignore_extra=True):
    load_model(net=net, file_path=get_model_file(model_name=model_name, local_model_store_dir_path=local_model_store_dir_path),
               ignore_extra=ignore_extra)


def calc_net_weight_count(net):
    """Number of trainable parameters (the quantity the reference's tests assert)."""
    return sum(p.numel() for p in net.parameters() if p.requires_grad)
