"""Command line entry with the reference's flags (main_rfn.py of the reference: same names, types and defaults), so
a job script written for the reference drives this implementation unchanged.  Additions (all optional):
  --synthetic_data   use the built-in SM-MNIST-shaped generator instead of files on disk (no dataset ships here);
  --max_steps        stop after that many optimizer steps.
Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N main_rfn.py ... --multigpu`; one process
per GPU, the global batch is sharded over ranks and gradients are all-reduced with RCCL (see rfn_hip/dist.py).
"""
import argparse
import os

# ROCm 7.2: with graph packet capture on, hipGraph memset nodes (PyTorch multi-block reductions zero their semaphores
# with one) race with neighbouring kernel nodes on replay; must be set before the HIP runtime initialises.
import sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
if "torch" not in sys.modules:  # the flag certainly precedes the HIP runtime: tell rfn_hip.graph_capture_safe()
    os.environ.setdefault("RFN_GRAPH_ENV_BEFORE_TORCH", "1")



def add_bool_arg(parser, name, help, default=False):
    group = parser.add_mutually_exclusive_group(required=False)
    group.add_argument("--" + name, dest=name, action="store_true", help=help)
    group.add_argument("--no-" + name, dest=name, action="store_false", help=help)
    parser.set_defaults(**{name: default})


def restricted_float(x):
    try:
        x = float(x)
    except ValueError:
        raise argparse.ArgumentTypeError("%r not a floating-point literal" % (x,))
    if x < 0.0 or x > 1.0:
        raise argparse.ArgumentTypeError("%r not in range [0.0, 1.0]" % (x,))
    return x


def convert_mixed_list(x):
    return int(x) if x.isdigit() else x


def convert_to_upscaler(x):
    return [convert_mixed_list(i) for i in x.split("-")]


def build_parser():
    p = argparse.ArgumentParser()
    # DATA
    p.add_argument("--batch_size", help="Specify batch size", default=35, type=int)
    p.add_argument("--n_frames", help="Specify number of frames", default=10, type=int)
    p.add_argument("--choose_data", help="Specify dataset", choices=["mnist", "bair", "kth"], default="bair", type=str)
    p.add_argument("--image_size", help="Specify the image size of mnist", default=64, type=int)
    p.add_argument("--digit_size", help="Specify the size of mnist digit", default=32, type=int)
    p.add_argument("--step_length", help="Specify the step size of mnist digit", default=4, type=int)
    p.add_argument("--num_digits", help="Specify the number of mnist digits", default=2, type=int)
    p.add_argument("--num_workers", help="Specify the number of workers in dataloaders", default=5, type=int)
    add_bool_arg(p, "use_validation_set", default=False, help="Specify if we want to use a validation set")
    # Trainer
    p.add_argument("--scheduler_type", help="Specify type of scheduler.", default="linear",
                   choices=["plateau", "linear"], type=str)
    p.add_argument("--patience_es", help="Specify patience for early stopping", default=50000000, type=int)
    p.add_argument("--patience_lr", help="Specify patience for lr_scheduler", default=10000000, type=int)
    p.add_argument("--factor_lr", help="Specify lr_scheduler factor (0..1)", default=.9999, type=restricted_float)
    p.add_argument("--min_lr", help="Specify minimum lr for scheduler", default=0.00005, type=float)
    p.add_argument("--n_bits", help="Specify number of bits", default=8, type=int)
    p.add_argument("--n_epochs", help="Specify number of epochs", default=100000, type=int)
    add_bool_arg(p, "verbose", default=False, help="Specify verbose mode (boolean)")
    p.add_argument("--path", help="Specify path to experiment", default="/content/", type=str)
    p.add_argument("--learning_rate", help="Specify learning_rate", default=0.0001, type=float)
    p.add_argument("--preprocess_range", help="Specify the range of the data for preprocessing",
                   choices=["0.5", "1.0"], default="0.5", type=str)
    p.add_argument("--preprocess_scale", help="Specify the scale for preprocessing", default=255, type=int)
    p.add_argument("--beta_max", help="Specify the maximum value of beta", default=1, type=float)
    p.add_argument("--beta_min", help="Specify the minimum value of beta", default=0.0000001, type=float)
    p.add_argument("--beta_steps", help="Specify the annealing steps", default=12000, type=int)
    p.add_argument("--n_predictions", help="Specify number of predictions", default=7, type=int)
    p.add_argument("--n_conditions", help="Specify number of predictions", default=3, type=int)
    add_bool_arg(p, "multigpu", default=False, help="Specify if we want to use multi GPUs")
    add_bool_arg(p, "load_model", default=False, help="Specify if we want to load a pre-existing model (boolean)")
    p.add_argument("--norm_type_features", help="Specify normalization type of layers upscaler/downscaler",
                   default="batchnorm", choices=["instancenorm", "batchnorm", "none"], type=str)
    # RFN
    p.add_argument("--x_dim", nargs="+", help="Specify data dimensions (b,c,h,w)", default=[32, 3, 64, 64], type=int)
    p.add_argument("--condition_dim", nargs="+", help="Specify condition dimensions (b,c,h,w)",
                   default=[32, 3, 64, 64], type=int)
    p.add_argument("--h_dim", help="Specify hidden state (h) channels", default=256, type=int)
    p.add_argument("--z_dim", help="Specify latent (z) channels", default=5, type=int)
    p.add_argument("--L", help="Specify flow depth", default=5, type=int)
    p.add_argument("--K", help="Specify flow recursion", default=15, type=int)
    p.add_argument("--extractor_structure", nargs="+",
                   help="Specify structure of extractor example writing, 32-32-conv 32-32-pool, creates 2 blocks",
                   default=[[8, 8, "pool", 16], [16, 16, "pool", 32], [32, 32, "pool", 64], [64, "pool", 128],
                            [128, "pool", 256]], type=convert_to_upscaler)
    p.add_argument("--norm_type", help="Specify normalization type of layers", default="none",
                   choices=["instancenorm", "batchnorm", "none"], type=str)
    p.add_argument("--upscaler_structure", nargs="+",
                   help="Specify upscaler structure, example writing, 32-32-deconv 32-32-upsample, creates 2 blocks",
                   default=[[256, 128], ["upsample", 128, 128], ["upsample", 64, 64], ["upsample", 32, 32],
                            ["upsample", 16, 16]], type=convert_to_upscaler)
    p.add_argument("--structure_scaler", help="Specify down/up-sampling channel factor", default=2, type=int)
    p.add_argument("--temperature", help="Specify temperature", default=0.7, type=restricted_float)
    p.add_argument("--prior_structure", help="Specify the structure of the prior", nargs="+", default=[256, 64],
                   type=convert_mixed_list)
    p.add_argument("--encoder_structure", help="Specify the structure of the encoder", nargs="+", default=[256, 64],
                   type=convert_mixed_list)
    p.add_argument("--skip_connection_flow", help="Specify skip_connections mode", default="with_skip",
                   choices=["without_skip", "with_skip", "only_skip"], type=str)
    add_bool_arg(p, "downscaler_tanh", default=False,
                 help="Specify if skip connection from downscaler is tanh'ed (boolean)")
    add_bool_arg(p, "upscaler_tanh", default=False, help="Specify if the outputs from the upscaler is tanh'ed (boolean)")
    add_bool_arg(p, "skip_connection_features", default=True,
                 help="Specify if skip connection between up and downscaler (boolean)")
    p.add_argument("--free_bits", help="Specify free bit, if -1.0 then we use no free_bit", default=-1.0, type=float)
    # Glow
    add_bool_arg(p, "learn_prior", default=True, help="Specify if we want a learned prior (boolean)")
    add_bool_arg(p, "LU_decomposed", default=True, help="Specify if we want to use LU factorization (boolean)")
    p.add_argument("--n_units_affine", help="Specify hidden units in affine coupling", default=256, type=int)
    p.add_argument("--non_lin_glow", help="Specify activation in glow", default="relu",
                   choices=["relu", "leakyrelu"], type=str)
    p.add_argument("--n_units_prior", help="Specify hidden units in prior", default=512, type=int)
    add_bool_arg(p, "make_conditional", default=True, help="Specify if split should be conditional or not (boolean)")
    p.add_argument("--flow_norm", help="Specify normalization type of glow-step", default="actnorm",
                   choices=["batchnorm", "actnorm"], type=str)
    p.add_argument("--base_norm", help="Specify normalization type of base distribution", default="actnorm",
                   choices=["batchnorm", "actnorm"], type=str)
    p.add_argument("--flow_batchnorm_momentum", help="Running average batchnorm momentum for flow-step", default=0.0,
                   type=float)
    p.add_argument("--clamp_type", help="Specify clamp type of affine coupling", default="realnvp",
                   choices=["glow", "realnvp", "softclamp", "none"], type=str)
    p.add_argument("--split2d_act", help="Specify clamp type of split2d", default="softplus",
                   choices=["softplus", "exp"], type=str)
    # overshooting, smoothing and resq
    p.add_argument("--a_dim", help="Channels of smoothing", default=200, type=int)
    add_bool_arg(p, "enable_smoothing", default=False, help="Enables smoothing")
    add_bool_arg(p, "res_q", default=False, help="Enables res_q")
    p.add_argument("--D", help="Number of overshoots", default=0, type=int)
    p.add_argument("--overshot_w", help="Weighting of overshooting", default=1.0, type=float)
    # additions of this implementation
    add_bool_arg(p, "synthetic_data", default=False, help="SM-MNIST-shaped synthetic video instead of files on disk")
    p.add_argument("--max_steps", help="Stop after this many optimizer steps (0 = no limit)", default=0, type=int)
    return p


def canonical_smmnist_argv(batch_size=32, n_frames=20):
    """The only concrete SM-MNIST configuration in the reference (RFN/default_rfn_job.sh:31-85) as an argv list."""
    b = str(batch_size)
    return ("--extractor_structure 16-16-pool-32 32-pool-64 64-pool-128 128-pool-256 256-pool-512 "
            "--upscaler_structure 256 upsample-128-128 upsample-64-64 upsample-32-32 upsample-16-16 "
            "--prior_structure 256 256 --encoder_structure 256 256 --make_conditional --learn_prior "
            "--skip_connection_features --flow_norm actnorm --step_length 4 --structure_scaler 2 --choose_data mnist "
            "--n_units_affine 256 --n_units_prior 512 --temperature 0.7 --norm_type none --z_dim 56 --h_dim 200 "
            "--beta_min 0.0001 --beta_steps 10000 --beta_max 1.0 --learning_rate 0.0001 --n_bits 8 "
            "--n_frames %d --image_size 64 --digit_size 28 --num_digits 2 --K 10 --L 5 "
            "--x_dim %s 1 64 64 --condition_dim %s 1 64 64 --batch_size %s --num_workers 4 --patience_lr 50 "
            "--skip_connection_flow without_skip --no-upscaler_tanh --no-downscaler_tanh --synthetic_data"
            % (n_frames, b, b, b)).split()


def main(args):
    from RFN.trainer import Solver
    if args.load_model:
        # rfn.pt may come from the reference (drop-in checkpoints): nothing from the file is executed
        ckpt = Solver.read_checkpoint("." + args.path + "model_folder/rfn.pt")
        # (a data-parallel checkpoint holds the global batch; this run may use another number of ranks)
        args = Solver.args_for_world(ckpt, int(os.environ.get("WORLD_SIZE", 1)) if ckpt["args"].multigpu else 1)
        solver = Solver(args)
        solver.build()
        solver.load(ckpt)
    else:
        solver = Solver(args)
        solver.build()
    solver.train()


if __name__ == "__main__":
    main(build_parser().parse_args())
