"""
Checkpoints in the reference's on-disk format (src/utils/Logger.py:36-46: a torch `.tar` with `decoder_state_dict`,
`gt_c2w_list`, `estimate_c2w_list`, `keyframe_list`, `idx`, `tracking_rendered_weight_list`, `addtional_map_records`), plus the
two hash tables (`hash_grid_sdf`, `hash_grid_color`: flat fp32 `.params`), which the reference does not save -- without them a
run cannot be resumed or re-rendered.  A reference checkpoint loads (tables stay as they are); ours loads
in the reference's eval tools, which read the keys they know and ignore the rest.
"""
import os

import torch


def save_checkpoint(path, slam, idx, tracking_rendered_weight_list=None, addtional_map_records=None):
    """slam: unislam_amd.slam.SLAM (or any object with decoders, es, ec, gt_c2w_list, estimate_c2w_list, mapper.keyframe_list)"""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    cpu = lambda t: t.detach().cpu().clone()
    torch.save({
        "decoder_state_dict": {k: cpu(v) for k, v in slam.decoders.state_dict().items()},
        "gt_c2w_list": cpu(slam.gt_c2w_list),
        "estimate_c2w_list": cpu(slam.estimate_c2w_list),
        "keyframe_list": list(slam.mapper.keyframe_list),
        "idx": int(idx),
        "tracking_rendered_weight_list": tracking_rendered_weight_list,
        "addtional_map_records": addtional_map_records,
        "hash_grid_sdf": cpu(slam.es.params), "hash_grid_color": cpu(slam.ec.params),
        "hash_grid_config": {"sdf": dict(slam.es.encoding_config), "color": dict(slam.ec.encoding_config)},
    }, path, _use_new_zipfile_serialization=False)
    return path


def load_checkpoint(path, decoders, hash_grid_sdf=None, hash_grid_color=None, map_location="cpu", trusted=False):
    """
    Restores the decoders (Tracker.py:254-style `load_state_dict`) and, when present, the tables IN PLACE (the parameters may be
    views of MapStep's flat buffer).  Returns the checkpoint dict (poses, keyframe list, idx).
    Everything save_checkpoint writes (tensors, lists, ints, plain dicts) loads under torch's weights_only unpickler, which is the
    default here: a `.tar` from an unknown source cannot run code.  trusted=True falls back to the full unpickler, which a
    REFERENCE checkpoint carrying arbitrary objects in tracking_rendered_weight_list / addtional_map_records needs (the
    reference itself always loads that way).
    """
    ck = torch.load(path, map_location=map_location, weights_only=not trusted)
    decoders.load_state_dict(ck["decoder_state_dict"])
    with torch.no_grad():
        for key, enc in (("hash_grid_sdf", hash_grid_sdf), ("hash_grid_color", hash_grid_color)):
            if enc is not None and key in ck:
                if ck[key].numel() != enc.params.numel():
                    raise ValueError(f"{key}: checkpoint has {ck[key].numel()} table parameters, the encoder {enc.params.numel()}")
                enc.params.copy_(ck[key].to(enc.params.device))
    return ck
