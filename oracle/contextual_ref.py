"""TEST INFRASTRUCTURE (oracle).  `ContextualLoss` of the reference's loss module, restated from the only artefact the
reference ships of it: /root/reference/__pycache__/loss.cpython-36.pyc (original loss.py:23-73; disassembly in SURVEY
Appendix B; names / defaults / line numbers pinned by tests/golden/loss_pyc_constants.json, extracted from that bytecode
by oracle/pin_loss_pyc.py).  PyTorch-CPU fp32, every step as the bytecode orders it."""
import torch
import torch.nn as nn


class ContextualLoss(nn.Module):
    def __init__(self, sigma=0.1, b=1.0, epsilon=1e-5, similarity='cos'):        # loss.py:24-29
        super().__init__()
        self.sigma, self.similarity, self.b, self.e = sigma, similarity, b, epsilon

    def cos_similarity(self, image_features, target_features):                     # loss.py:31-44
        B, C = image_features.size(0), image_features.size(1)
        i = image_features.view(B, C, -1).permute(0, 2, 1)
        t = target_features.view(B, C, -1).permute(0, 2, 1)
        mu = torch.mean(t, 1, True)
        ic, tc = i - mu, t - mu
        il = torch.div(ic, torch.sqrt(torch.sum(ic * ic, dim=2, keepdim=True)))
        tl = torch.div(tc, torch.sqrt(torch.sum(tc * tc, dim=2, keepdim=True)))
        return 1 - torch.bmm(il, tl.permute(0, 2, 1))

    def L2_similarity(self, image_features, target_features):                      # loss.py:46-47 (a stub)
        pass

    def relative_distances(self, distances):                                       # loss.py:49-51
        return distances / (torch.min(distances, dim=2, keepdim=True)[0] + self.e)

    def weighted_average_distances(self, distances_normalized):                    # loss.py:53-57
        w = torch.exp((self.b - distances_normalized) / self.sigma)
        return torch.div(w, torch.sum(w, dim=2, keepdim=True))

    def CX(self, distances):                                                       # loss.py:59-68
        cx = self.weighted_average_distances(self.relative_distances(distances))
        m = torch.max(cx.permute(0, 2, 1), dim=1)[0]
        cs = torch.mean(m, dim=1)
        return torch.mean(-torch.log(cs))

    def forward(self, image_features, target_features):                            # loss.py:70-73
        if self.similarity != 'cos':
            raise NotImplementedError("only the cosine similarity is implemented in the reference (L2_similarity is `pass`)")
        return self.CX(self.cos_similarity(image_features, target_features))
