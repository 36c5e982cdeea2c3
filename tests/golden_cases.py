"""Golden-vector case table shared by tools/capture_golden.py (which runs the *reference*
classes in the build container) and the tests (which run the oracle / the HIP path).

Each case: reference class name, ModelConfig overrides, input spec, forward kwargs.
Input spec entries are (T, ...) per modality; T == 0 means a 2-D (B, d) tensor.
"""
CASES = {
    # BASELINE.json configs[0]: early-fusion plumbing, B=4, d=256
    "early_c1": dict(cls="EarlyFusion", cfg=dict(fusion_hidden_size=256), B=4, Ts=(0, 0, 0)),
    "late": dict(cls="LateFusion", cfg=dict(fusion_hidden_size=192), B=4, Ts=(0, 0, 0)),
    # one cross block, multi-tile keys (Tk > 64) and ragged tails, head_dim 96
    "cross_dh96": dict(cls="CrossModalTransformer", cfg=dict(fusion_hidden_size=192, fusion_num_heads=2),
                       B=3, Ts=(70, 40), two_inputs=True),
    "cross_dh64_long": dict(cls="CrossModalTransformer", cfg=dict(fusion_hidden_size=128, fusion_num_heads=2),
                            B=2, Ts=(37, 150), two_inputs=True),
    # MulT with unequal sequence lengths (the BASELINE (B,T,d) route), head_dim 96 and 64
    "mult_seq_dh96": dict(cls="MultimodalTransformer", cfg=dict(fusion_hidden_size=192, fusion_num_heads=2),
                          B=3, Ts=(70, 40, 6)),
    "mult_seq_dh64": dict(cls="MultimodalTransformer", cfg=dict(fusion_hidden_size=128, fusion_num_heads=2),
                          B=2, Ts=(33, 65, 5)),
    # several 128-row query tiles and many 32-key blocks per head (the online-softmax path of the kernels), ragged tails
    "mult_seq_long": dict(cls="MultimodalTransformer", cfg=dict(fusion_hidden_size=192, fusion_num_heads=2),
                          B=2, Ts=(300, 130, 30)),
    # the BASELINE width and head count (d=768, 8 heads of 96) at short lengths
    "mult_seq_d768": dict(cls="MultimodalTransformer", cfg=dict(fusion_hidden_size=768, fusion_num_heads=8),
                          B=2, Ts=(40, 24, 6)),
    # MulT as wired by the reference model: pooled (B,d) features, T = 1
    "mult_2d": dict(cls="MultimodalTransformer", cfg=dict(fusion_hidden_size=192, fusion_num_heads=2),
                    B=4, Ts=(0, 0, 0)),
    "contrastive": dict(cls="ContrastiveFusion", cfg=dict(fusion_hidden_size=192), B=5, Ts=(0, 0, 0),
                        kwargs=dict(compute_contrastive_loss=True)),
    "adaptive": dict(cls="AdaptiveFusion", cfg=dict(fusion_hidden_size=192, fusion_num_heads=2),
                     B=4, Ts=(0, 0, 0)),
    # graph + hierarchical: composition pinned with a sparse edge-list GAT stand-in,
    # GAT arithmetic itself PARITY UNPINNED (torch_geometric not installable offline)
    "graph": dict(cls="GraphFusion", cfg=dict(fusion_hidden_size=128, graph_hidden_size=128,
                                              graph_num_layers=2), B=3, Ts=(0, 0, 0), gat_unpinned=True),
    "hier_ref": dict(cls="HierarchicalFusion",
                     cfg=dict(fusion_hidden_size=128, fusion_num_heads=2, graph_hidden_size=128,
                              graph_num_layers=2), B=4, Ts=(0, 0, 0),
                     kwargs=dict(compute_contrastive_loss=True), gat_unpinned=True),
    "adapter": dict(cls="AdapterLayer", module="encoders", ctor=(96, 16), B=3, Ts=(7,), d=96),
}

COMMON_CFG = dict(fusion_dropout=0.0, graph_dropout=0.0)
SMALL_GRAD_NUMEL = 4096      # full gradients are stored for parameters up to this size
