"""Teacher-forced generation fine-tune on MI355X: counterpart of the reference's `GenerationTrainer`
(finetune_generation.py:57-290) -- SURVEY 8(f-2), the first "next" row after the pre-train path.

Same forward/backward as the pre-train step with a different mask and weights:
  * decoder input = the encoder input itself (`y_shift = x`, finetune_generation.py:155), masks = bar column != PAD;
  * per-head CE averaged over the decoder attention mask, head weights 0.3 (Instrument, TimeSig, Tempo) / 1.5 (Pitch) / 1,
    times len(e2w[etype]) in dict order, divided by sum(n_tok) (finetune_generation.py:236-250);
  * accuracy over the same mask (finetune_generation.py:188-193); argmax ids collected in test mode.
The shape-similarity "FAD" metrics (finetune_generation.py:185-223) need the third-party `shapesimilarity` package, which is
not installed in this image: they are reported as 0.0 (host-side metric, outside the hot path).
"""
import shutil
import sys

import numpy as np
import torch

from . import ops
from ._lib import PBError
from .model import PianoBartLM

HEAD_WEIGHT = [1.0, 1.0, 0.3, 1.5, 1.0, 1.0, 0.3, 0.3]          # finetune_generation.py:241-248 (index = head i)


class GenerationTrainer:
    def __init__(self, pianobart, train_dataloader, valid_dataloader, test_dataloader, lr, testset_shape, cpu, cuda_devices=None, model=None):
        if cpu or not torch.cuda.is_available():
            raise PBError('pianobart_amd has no CPU execution path')
        if cuda_devices is not None and len(cuda_devices) > 1:
            raise PBError('nn.DataParallel is replaced by one process per GPU (torch.distributed.run)')
        self.device = torch.device('cuda', cuda_devices[0] if cuda_devices else 0)
        print('   device:', self.device)
        self.pianobart = pianobart
        self.model = (model if model is not None else PianoBartLM(pianobart)).to(self.device)
        self.engine = self.model._get_engine()
        self.engine.bind(self.device)
        self.train_data, self.valid_data, self.test_data = train_dataloader, valid_dataloader, test_dataloader
        self.testset_shape = testset_shape
        self.lr = lr
        n_tok = np.array([len(pianobart.e2w[k]) for k in pianobart.e2w], dtype=np.float64)       # dict order, as the reference
        w = np.array(HEAD_WEIGHT) * n_tok
        self._hw = torch.tensor(w, dtype=torch.float32, device=self.device)
        self._scale = float(w.sum() / n_tok.sum())
        self._wnp, self._ntok = np.array(HEAD_WEIGHT), n_tok

    def train(self):
        self.model.train()
        return self.iteration(self.train_data, 0)

    def valid(self):
        self.model.eval()
        return self.iteration(self.valid_data, 1)

    def test(self):
        self.model.eval()
        return self.iteration(self.test_data, 2)

    def iteration(self, training_data, mode):
        eng, pad = self.engine, int(self.pianobart.bar_pad_word)
        total_acc, total_loss = np.zeros(8), 0.0
        all_output, cnt = (torch.empty(self.testset_shape) if mode == 2 else None), 0
        for x, y in training_data:
            x, y = x.to(self.device).long(), y.to(self.device).long()
            B, S = x.shape[:2]
            x16, y16 = ops.ids_to_i16(x), ops.ids_to_i16(y)
            attn_enc = (x16[:, :, 0] != pad).float()
            attn_dec = attn_enc                                        # y_shift = x
            loss_mask = attn_dec[:, :, None].expand(B, S, 8).contiguous()
            am = torch.empty(B * S, 8, dtype=torch.int16, device=self.device) if mode == 2 else None
            sums = eng.loss_and_grads(x16, x16, y16, loss_mask, attn_enc, attn_dec, train=(mode == 0), head_w=self._hw, w_scale=self._scale,
                                      argmax_out=am)
            if mode == 0:
                eng.optimizer_step(lr=self.lr)
            s = sums.double().cpu().numpy()
            losses = s[0:8] / s[8:16] * self._wnp                      # the reference logs the weighted per-head losses
            accs = s[16:24] / s[8:16]
            loss = float((losses * self._ntok).sum() / self._ntok.sum())
            if mode == 2:
                all_output[cnt:cnt + B] = am.view(B, S, 8).float().cpu()
                cnt += B
            sys.stdout.write('Loss: {:06f} | loss: {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}\n'.format(loss, *losses))
            sys.stdout.write('Acc: {:06f} | acc: {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}, {:03f}\n'.format(np.average(accs), *accs))
            sys.stdout.write('FAD(BAR) Similarity: {:0.6f} , FAD Similarity {:0.6f} \n'.format(0.0, 0.0))
            total_acc += accs
            total_loss += loss
        n = max(1, len(training_data))
        out = (round(total_loss / n, 4), [round(float(a) / n, 4) for a in total_acc], 0.0, 0.0)
        return out + (all_output,) if mode == 2 else out

    def save_checkpoint(self, epoch, train_acc, valid_acc, valid_loss, train_loss, is_best, filename):
        """finetune_generation.py:276-290: whole-model state_dict (pianobart.* + mask_lm.*)."""
        eng = self.engine
        state = {'epoch': epoch + 1, 'state_dict': {k: v.detach().cpu() for k, v in self.model.state_dict().items()}, 'valid_acc': valid_acc,
                 'valid_loss': valid_loss, 'train_loss': train_loss, 'train_acc': train_acc,
                 'optimizer': {'step': eng.step_count, 'lr': self.lr, 'exp_avg': None if eng.opt_m is None else eng.opt_m.cpu(),
                               'exp_avg_sq': None if eng.opt_v is None else eng.opt_v.cpu()}}
        torch.save(state, filename)
        if is_best:
            shutil.copyfile(filename, filename.split('.')[0] + '_best.ckpt')
