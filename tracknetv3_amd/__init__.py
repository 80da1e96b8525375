"""tracknetv3_amd -- MI355X-native hot path of TrackNetV3 (see DESIGN.md / INTEGRATION.md).

    from tracknetv3_amd.utils.general import get_model          # TrackNet / InpaintNet with the reference's state_dict layout
    from tracknetv3_amd.utils.metric import WBCELoss
    from tracknetv3_amd.postprocess import predict, predict_location, get_ensemble_weight, generate_inpaint_mask
    from tracknetv3_amd.pipeline import predict_video           # predict.py-shaped end-to-end driver
    from tracknetv3_amd.parallel import TrackNetTrainer         # train.py hot loop (+ RCCL data parallelism)
"""
__version__ = "0.1.0"
