"""Evaluator — the bits-per-dim part of the reference's evaluation driver (evaluation_metrics/error_metrics.py): the
`compute_loss` bookkeeping (:358-368) and the test-set loop `get_loss` (:370-417) that re-uses `Solver.preprocess` and
`RFN.loss`, plus thin wrappers over the model's analysis methods (`RFN.reconstruct_elbo_gap`, `.probability_future`,
`.param_analysis`, RFN/RFN_new.py:496-788).  The image-quality metrics of that file (PSNR / SSIM / LPIPS / FVD) and its
plotting need lpips, skimage and a TF-hub network and are outside the hot-path scope."""
import numpy as np
import torch


class Evaluator(object):
    def __init__(self, solver, args=None, settings=None):
        self.solver = solver
        self.model = solver.model
        self.args = args if args is not None else solver.args
        self.choose_data = getattr(self.args, "choose_data", "mnist")
        self.test_loader = getattr(solver, "test_loader", None)
        # the reference evaluates the loss on as many frames as the model was trained on (:387-388)
        self.n_trained = getattr(settings, "n_trained", None) or getattr(self.args, "n_frames", None)
        self.device = solver.device

    def compute_loss(self, nll, kl, dims, t=10):
        """error_metrics.py:358-368 -> (bits/dim, kl / t, nll / t)"""
        kl_store, nll_store = kl.detach(), nll.detach()
        elbo = -(kl_store + nll_store)
        bits_per_dim_loss = float(-elbo / (np.log(2.) * torch.prod(torch.tensor(dims)) * t))
        return bits_per_dim_loss, float(kl_store / t), float(nll_store / t)

    def get_loss(self, model_name="rfn.pt", loss_resamples=1, loader=None, max_batches=None):
        """error_metrics.py:370-417: mean (and, with resampling, standard deviation) of the per-batch bits/dim over the
        test set, model in eval mode.  Only the RFN branch exists here (the reference's other branch is the
        importance-weighted bound of its VRNN / SRNN baselines)."""
        assert model_name == "rfn.pt", "only the RFN loss is on the hot path"
        loader = loader if loader is not None else self.test_loader
        with torch.no_grad():
            self.model.eval()
            means = []
            for _ in range(loss_resamples):
                bpd = []
                for batch_i, true_image in enumerate(loader):
                    if max_batches is not None and batch_i >= max_batches:
                        break
                    image = true_image[0] if self.choose_data == "bair" and isinstance(true_image, (list, tuple)) else true_image
                    image = self.solver.preprocess(image.to(self.device))
                    imageloss = image[:, :self.n_trained] if self.n_trained else image
                    _, kl, nll = self.model.loss(imageloss, 0)
                    b, _, _ = self.compute_loss(nll=nll, kl=kl, dims=imageloss.shape[2:], t=imageloss.shape[1] - 1)
                    bpd.append(b)
                means.append(torch.FloatTensor(bpd))
            means = torch.stack(means)
            mean = means.mean()
            std = means.std() if loss_resamples > 1 else -1
        return mean, std

    # ---- the analyses the reference's evaluator drives (error_metrics.py: plot_elbo_gap, plot_prob_of_t, param_plots)
    def elbo_gap(self, image, sample=False):
        self.model.eval()
        return self.model.reconstruct_elbo_gap(self.solver.preprocess(image.to(self.device)), sample=sample)

    def probability_future(self, image, n_conditions):
        self.model.eval()
        return self.model.probability_future(self.solver.preprocess(image.to(self.device)), n_conditions)

    def param_analysis(self, image, n_predictions, n_conditions):
        self.model.eval()
        return self.model.param_analysis(self.solver.preprocess(image.to(self.device)), n_predictions, n_conditions)
