"""Command-line surface of the reference (``/root/reference/main.py:9-147``): the same 119 flags
with the same names, types and defaults, table-driven.  Flags whose features are outside the
MI355X hot path are accepted here and rejected when the model is built (``model.BigGAN.__init__``).

    python -m biggan_tensorflow_amd.main --phase train --gan_type hinge --img_size 128 --ch 64 --batch_size 64
"""
import argparse

from .utils import check_folder, str2bool

B, I, F, T = str2bool, int, float, str

# (name, type, default)  -- order follows main.py:9-147
FLAGS = [
    ("phase", T, "train"), ("dataset", T, "celebA-HQ"),
    ("epoch", I, 50), ("iteration", I, 10000), ("batch_size", I, 16), ("virtual_batches", I, 1),
    ("ch", I, 64), ("d_ch", I, 0), ("deep", B, False),
    ("print_freq", I, 250), ("save_freq", I, 1000), ("histogram_freq", I, 125), ("keep_checkpoints", I, 5),
    ("g_lr", F, 0.00005), ("d_lr", F, 0.0002),
    ("beta1", F, 0.0), ("beta2", F, 0.9), ("moving_decay", F, 0.999),
    ("z_dim", I, 256), ("shared_z", I, 0), ("c_dim", I, 3), ("alpha_mask", B, True), ("g_alpha_helper", B, True),
    ("first_split_ratio", I, 3), ("z_reconstruct", B, False), ("sn", B, True), ("bn_in_d", B, False),
    ("bias_in_d", B, False), ("bias_in_sa", B, True), ("bn_type", T, "batch_norm"), ("bn_momentum", F, 0.98),
    ("bn_renorm_rmax", F, 1.5), ("bn_renorm_dmax", F, 0.5), ("bn_renorm_momentum", F, 0.9),
    ("bn_renorm_shared", B, False), ("g_regularization", T, "ortho_cosine"), ("g_regularization_factor", F, 0.0001),
    ("conv_padding", T, "reflect"), ("upsampling_method", T, "deconv4"), ("downsampling_method", T, "strided_conv3"),
    ("g_conv", T, "deconv3"), ("g_grow_factor", F, 2.0), ("d_grow_factor", F, 2.0),
    ("g_sa_size", I, 0), ("d_sa_size", I, 0), ("sa_size", I, 0),
    ("gan_type", T, "ra-dragan"), ("d_loss_func", T, ""), ("activation", T, "prelu"), ("ld", F, 10.0),
    ("multi_head", B, False), ("d_flood", F, 0.1), ("g_flood", F, 0.05),
    ("n_critic", I, 1),
    ("img_size", I, 256), ("sample_num", I, 64), ("static_sample_z", B, True), ("static_sample_seed", I, 123456789),
    ("z_trunc_train", B, True), ("z_trunc_sample", B, True), ("save_morphs", B, False), ("sample_ema", T, "ema"),
    ("random_flip", B, True), ("da_policy", T, "full"),
    ("n_labels", I, 0), ("cls_embedding", B, False), ("cls_embedding_size", I, 0), ("cls_embedding_concat", B, False),
    ("label_file", T, ""), ("ignore_missing_labels", B, False), ("cls_loss_type", T, "logistic"),
    ("weight_file", T, ""), ("g_first_level_dense_layer", B, True), ("g_other_level_dense_layer", B, False),
    ("g_no_last_resblock", B, False), ("g_z_dense_concat", B, False), ("d_cls_dense_layers", B, False),
    ("d_compat_use_sn_in_classification", B, False), ("d_compat_use_sn_in_critic_output", B, True),
    ("g_mixed_resblocks", B, False), ("g_mixed_resblock_ch_div", F, 2.0), ("g_final_layer", B, False),
    ("g_final_layer_extra", B, False), ("g_final_layer_extra_bias", B, False), ("g_final_kernel", T, "3"),
    ("g_final_kernel_extra", T, "3"), ("g_final_layer_shortcuts", B, False), ("g_final_layer_shortcuts_after", I, 0),
    ("g_final_mixed_conv", B, False), ("g_final_mixed_conv_stacks", I, 2), ("g_final_mixed_conv_z_layers", T, "none"),
    ("g_final_mixed_nodeconv2", B, False), ("g_rgb_mix_kernel", I, 3), ("d_cls_loss_weight", F, 5.0),
    ("g_cls_loss_weight", F, 1.0), ("save_cls_samples", B, False), ("cls_loss_weights", T, ""),
    ("save_cls_samples_to", T, ""), ("load_cls_samples_from", T, ""), ("d_reconstruction", B, False),
    ("d_reconstruction_halfres", B, False), ("d_reconstruction_texture", B, False), ("d_tex_recon_feat_size", I, 16),
    ("d_tex_recon_patch_div", I, 4), ("d_recon_ch", I, 64), ("d_tex_recon_ch", I, 96), ("d_recon_ld", F, 1.0),
    ("d_tex_recon_ld", F, 0.5), ("d_recon_bn_after_act", B, False), ("d_save_recon_samples", B, False),
    ("d_final_conv", B, False),
    ("test_num", I, 10), ("allow_growth", B, False),
    ("checkpoint", T, ""), ("checkpoint_dir", T, "checkpoint"), ("result_dir", T, "results"), ("log_dir", T, "logs"),
    ("sample_dir", T, "samples"), ("request_dir", T, "request"),
]


# Extensions of this build (not flags of the reference, kept out of FLAGS):
#   --precision fp32        the reference's arithmetic (ops.py:14): fp32 tensors, fp32 MFMA          [default]
#               bf16-staged fp32 tensors, conv / large GEMM operands rounded to bf16 while staged, fp32 accumulate
#               bf16        BASELINE configs 3-5: bf16-resident activations + packed bf16 weights, fp32 accumulate,
#                           fp32 master weights / optimiser / statistics
#   the default can also be given by the environment variable BIGGAN_PRECISION
EXTRA_FLAGS = [("precision", T, None)]


def build_parser():
    parser = argparse.ArgumentParser(description="MI355X-native BigGAN training step (flag surface of the reference)")
    for name, typ, default in FLAGS + EXTRA_FLAGS:
        parser.add_argument("--" + name, type=typ, default=default)
    return parser


def check_args(args, make_dirs=True):
    """main.py:152-176: creates the four output folders; the two sanity checks only print."""
    if make_dirs:
        for d in (args.checkpoint_dir, args.result_dir, args.log_dir, args.sample_dir):
            check_folder(d)
    if not args.epoch >= 1:
        print('number of epochs must be larger than or equal to one')
    if not args.batch_size >= 1:
        print('batch size must be larger than or equal to one')
    return args


def parse_args(argv=None, make_dirs=True):
    import os
    args = build_parser().parse_args(argv)
    if args.precision is None:
        args.precision = os.environ.get("BIGGAN_PRECISION", "fp32")
    return check_args(args, make_dirs)


def main(argv=None):
    args = parse_args(argv)
    if args is None:
        exit()
    import torch
    from . import parallel
    from .model import BigGAN
    rank, world, local = parallel.init_from_env()        # torchrun: one process per GPU, RCCL
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    gan = BigGAN(args)
    gan.build_model()
    if args.phase == 'train':
        gan.train()
        print(" [*] Training finished!")
    elif args.phase == 'test':
        gan.test()
        print(" [*] Test finished!")
    else:
        raise NotImplementedError("--phase %s (the reference's HTTP sample service) is outside the MI355X hot path"
                                  % args.phase)


if __name__ == '__main__':
    main()
