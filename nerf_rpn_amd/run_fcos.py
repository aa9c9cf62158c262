"""Train / evaluate / benchmark the anchor-free (FCOS) NeRF-RPN on MI355X.

Command-line drop-in for the reference's ``nerf_rpn/run_fcos.py`` (flags, defaults and choices of run_fcos.py:30-131; side-effect
files ``<save_path>/{model_best.pt, epoch_N.pt, eval.json, proposals/*.npz, voxel_scores/*.npz}``; checkpoint keys ``epoch,
backbone_state_dict, fcos_state_dict, train_args``; proposal files hold ``proposals``, ``scores`` and, with
``--save_level_index``, ``level_indices``).  Underneath: HIP kernels, one process per GPU with ``engine.FlatTrainer`` instead of
DDP(find_unused_parameters) + torch.optim.  Extra flag: ``--dtype {fp32,bf16}``.
"""
import argparse
import json
import os

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import run_rpn
from .eval import evaluate_box_proposals_ap, evaluate_box_proposals_recall
from .model.fcos import FCOSOverNeRF
from .model.feature_extractor import VGG_FPN, Bottleneck, ResNet_FPN_256, SwinTransformer_FPN
from .model.utils import box_iou_3d


def build_parser():
    p = argparse.ArgumentParser(description='Train and eval the NeRF RPN baseline using FCOS.')
    p.add_argument('--mode', default='train', choices=['train', 'eval', 'benchmark'])
    p.add_argument('--dataset', '--dataset_name', default='hypersim', choices=['hypersim', 'front3d', 'general', 'scannet'])
    p.add_argument('--features_path', default='', help='The path to the features.')
    p.add_argument('--boxes_path', default='', help='The path to the boxes.')
    p.add_argument('--save_path', default='', help='The path to save the model.')
    p.add_argument('--dataset_split', default='', help='The dataset split to use.')
    p.add_argument('--checkpoint', default='', help='The path to the checkpoint to load.')
    p.add_argument('--load_backbone_only', action='store_true', help='Only load the backbone weights.')
    p.add_argument('--preload', action='store_true', help='Preload the features and boxes.')
    p.add_argument('--train_csv', default='', help='The path to the train csv.')
    p.add_argument('--val_csv', default='', help='The path to the val csv.')
    p.add_argument('--test_csv', default='', help='The path to the test csv.')
    p.add_argument('--backbone_type', type=str, default='swin_s', choices=['resnet', 'vgg_AF', 'vgg_EF', 'swin_t', 'swin_s', 'swin_b', 'swin_l'])
    p.add_argument('--input_dim', type=int, default=4, help='Input dimension for backbone.')
    p.add_argument('--rotated_bbox', action='store_true')
    p.add_argument('--resolution', type=int, default=160, help='The max resolution of the input features.')
    p.add_argument('--normalize_density', action='store_true', help='Whether to normalize the density.')
    p.add_argument('--output_proposals', action='store_true', help='Whether to output proposals during evaluation.')
    p.add_argument('--save_level_index', action='store_true', help='Whether to save level indices')
    p.add_argument('--filter', choices=['none', 'tp', 'fp'], default='none')
    p.add_argument('--filter_threshold', type=float, default=0.7)
    p.add_argument('--output_voxel_scores', action='store_true')
    p.add_argument('--batch_size', default=1, type=int, help='The batch size.')
    p.add_argument('--num_epochs', default=100, type=int, help='The number of epochs to train.')
    p.add_argument('--lr', default=1e-4, type=float, help='The learning rate.')
    p.add_argument('--reg_loss_weight', default=1.0, type=float)
    p.add_argument('--weight_decay', default=0.01, type=float)
    p.add_argument('--clip_grad_norm', default=0.1, type=float, help='The gradient clipping norm.')
    p.add_argument('--log_interval', default=20, type=int)
    p.add_argument('--log_to_file', action='store_true')
    p.add_argument('--eval_interval', default=1, type=int)
    p.add_argument('--keep_checkpoints', default=1, type=int)
    p.add_argument('--wandb', action='store_true')
    p.add_argument('--rotate_prob', default=0.5, type=float)
    p.add_argument('--flip_prob', default=0.5, type=float)
    p.add_argument('--rot_scale_prob', default=0.5, type=float)
    p.add_argument('--gpus', default='')
    p.add_argument('--num_convs', default=4, type=int)
    p.add_argument('--norm_reg_targets', action='store_true')
    p.add_argument('--centerness_on_reg', action='store_true')
    p.add_argument('--center_sampling_radius', default=1.5, type=float)
    p.add_argument('--iou_loss_type', choices=['iou', 'linear_iou', 'giou', 'diou', 'smooth_l1'], default='iou')
    p.add_argument('--use_additional_l1_loss', action='store_true')
    p.add_argument('--conv_at_start', action='store_true')
    p.add_argument('--proj2d_loss_weight', default=0.0, type=float)
    p.add_argument('--pre_nms_top_n', default=2500, type=int)
    p.add_argument('--fpn_post_nms_top_n', default=2500, type=int)
    p.add_argument('--nms_thresh', default=0.3, type=float)
    p.add_argument('--pre_nms_thresh', default=0.0, type=float)
    p.add_argument('--min_size', default=0.0, type=float)
    p.add_argument('--ap_top_n', default=None, type=int)
    p.add_argument('--output_all', action='store_true')
    p.add_argument('--dtype', choices=['fp32', 'bf16'], default=None, help='Compute dtype of the HIP conv path (default: bf16 for train/benchmark, fp32 for eval).')
    return p


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    args.dataset_name = args.dataset        # the shared Trainer plumbing reads the run_rpn.py spelling
    return args


class Trainer(run_rpn.Trainer):
    def __init__(self, args, rank=0, world_size=1, device_id=None, logger=None):
        import logging
        self.args, self.rank, self.world_size, self.device_id = args, rank, world_size, device_id
        self.logger = logger if logger is not None else logging.getLogger()
        self.dataset = {'hypersim': run_rpn.HypersimRPNDataset, 'front3d': run_rpn.Front3DRPNDataset, 'general': run_rpn.GeneralRPNDataset,
                        'scannet': run_rpn.ScanNetRPNDataset}[args.dataset]
        if args.wandb and rank == 0:
            import wandb
            wandb.init(project='nerf-rpn', config=dict(vars(args)))
        self.logger.info('Constructing model...')
        self.build_backbone()
        dtype = args.dtype or ('fp32' if args.mode == 'eval' else 'bf16')
        self.model = FCOSOverNeRF(args=args, backbone=self.backbone, fpn_strides=[4, 8, 16, 32], world_size=world_size,
                                  compute_dtype=torch.bfloat16 if dtype == 'bf16' else torch.float32)
        if args.checkpoint:
            assert os.path.exists(args.checkpoint), 'The checkpoint does not exist.'
            self.logger.info(f'Loading checkpoint from {args.checkpoint}.')
            ckpt = torch.load(args.checkpoint, map_location='cpu')
            self.model.backbone.load_state_dict(ckpt['backbone_state_dict'])
            if not args.load_backbone_only:
                self.model.fcos_module.load_state_dict(ckpt['fcos_state_dict'])
        self.num_bbox_digits = 7 if args.rotated_bbox else 6
        self.model.cuda()
        self.init_datasets()

    def build_backbone(self):
        t = self.args.backbone_type
        if self.args.input_dim != 4 or self.args.conv_at_start:
            raise NotImplementedError('the HIP backbones take the 4-channel rgb-sigma grid without extra starting convs')
        if t == 'resnet':
            self.backbone = ResNet_FPN_256(Bottleneck, [3, 4, 6, 3], input_dim=4, is_max_pool=True)
        elif t in ('vgg_AF', 'vgg_EF'):
            self.backbone = VGG_FPN(t[-2:], 4, True, self.args.resolution)
        else:       # run_fcos.py:207-219: stochastic depth 0 for FCOS
            swin = {'swin_t': (96, [2, 2, 6, 2], [3, 6, 12, 24]), 'swin_s': (96, [2, 2, 18, 2], [3, 6, 12, 24]),
                    'swin_b': (128, [2, 2, 18, 2], [3, 6, 12, 24]), 'swin_l': (192, [2, 2, 18, 2], [6, 12, 24, 48])}[t]
            self.backbone = SwinTransformer_FPN(patch_size=[4, 4, 4], embed_dim=swin[0], depths=swin[1], num_heads=swin[2],
                                                window_size=[4, 4, 4], stochastic_depth_prob=0, expand_dim=True)

    def save_checkpoint(self, epoch, path):
        torch.save({'epoch': epoch, 'backbone_state_dict': self.model.backbone.state_dict(),
                    'fcos_state_dict': self.model.fcos_module.state_dict(), 'train_args': self.args.__dict__}, path)

    def train_epoch(self, epoch):
        a = self.args
        for i, (rgbsigma, boxes, scene_name) in enumerate(self.train_loader):
            self.model.train()
            rgbsigma = self.scenes_to_device(rgbsigma)
            boxes = [t.cuda(non_blocking=True) for t in boxes]
            _, losses, _ = self.model(rgbsigma, boxes)
            lc, lr_, lt = losses['loss_cls'], losses['loss_reg'] * a.reg_loss_weight, losses['loss_centerness']
            loss = lc + lr_ + lt
            loss.backward()
            lr = self.trainer.step()
            if i % a.log_interval == 0:
                vals = self.trainer.reduce_scalars(loss, lc, lr_, lt).tolist()
                if self.rank == 0:
                    self.logger.info(f'epoch {epoch} [{i}/{len(self.train_loader)}]  lr: {lr:.6f}  loss: {vals[0]:.4f}  '
                                     f'loss_cls: {vals[1]:.6f}, loss_reg: {vals[2]:.6f}, loss_centerness: {vals[3]:.6f}')

    def output_proposals(self, scenes, proposals, scores, gt_boxes):
        out = os.path.join(self.args.save_path, 'proposals')
        os.makedirs(out, exist_ok=True)
        for scene, proposal, score, gt in zip(scenes, proposals, scores, gt_boxes):
            level_index = None
            if self.args.save_level_index:
                level_index, proposal = proposal[..., 0], proposal[..., 1:]
            if self.args.filter != 'none':
                if proposal.shape[0] == 0 or gt is None or gt.shape[0] == 0:
                    continue
                keep = box_iou_3d(gt.cuda(), proposal.cuda()).max(dim=0)[0].cpu() > self.args.filter_threshold
                if self.args.filter == 'fp':
                    keep = ~keep
                proposal, score = proposal[keep], score[keep]
                if level_index is not None:
                    level_index = level_index[keep]
            if level_index is not None:
                np.savez(os.path.join(out, f'{scene}.npz'), proposals=proposal, scores=score, level_indices=level_index)
            else:
                np.savez(os.path.join(out, f'{scene}.npz'), proposals=proposal, scores=score)

    @torch.no_grad()
    def eval(self, dataset):
        a = self.args
        self.model.eval()
        loader = DataLoader(dataset, batch_size=max(1, a.batch_size // self.world_size), shuffle=False, num_workers=4, collate_fn=dataset.collate_fn)
        self.logger.info('Evaluating...')
        proposals_list, scores_list, gt_list, scenes_list = [], [], [], []
        for rgbsigma, gt_boxes, scenes in loader:
            rgbsigma = self.scenes_to_device(rgbsigma)
            paths = None
            if a.output_voxel_scores:
                d = os.path.join(a.save_path, 'voxel_scores')
                os.makedirs(d, exist_ok=True)
                paths = [os.path.join(d, f'{s}.npz') for s in scenes]
            proposals, _, scores = self.model(rgbsigma, objectness_output_paths=paths)
            proposals_list += [p.cpu() for p in proposals]
            scores_list += [s.cpu() for s in scores]
            gt_list += [b.cpu() if b is not None else None for b in gt_boxes]
            scenes_list += list(scenes)
        if not a.save_level_index:
            proposals_list = [p[..., 1:] for p in proposals_list]
        if a.output_proposals:
            self.output_proposals(scenes_list, proposals_list, scores_list, gt_list)
        if a.save_level_index:
            proposals_list = [p[..., 1:] for p in proposals_list]
        if gt_list[0] is None:
            return None, None
        recalls, APs, js = [], [], {}
        for limit in [300, 1000, a.fpn_post_nms_top_n]:
            if limit > a.fpn_post_nms_top_n:
                continue
            r50 = evaluate_box_proposals_recall(proposals_list, scores_list, gt_list, thresholds=torch.tensor([0.5]), limit=limit)
            r25 = evaluate_box_proposals_recall(proposals_list, scores_list, gt_list, thresholds=torch.tensor([0.25]), limit=limit)
            ar = evaluate_box_proposals_recall(proposals_list, scores_list, gt_list, thresholds=torch.arange(0.25, 1.0, 0.05), limit=limit)
            recalls.append(r50['ar'].item())
            js[f'recall_50_top_{limit}'], js[f'recall_25_top_{limit}'], js[f'recall_ar_top_{limit}'] = r50, r25, ar
            print(f'\nTop {limit} proposals:\nRecall@50: Recall: {r50["ar"].item():.4f}, Num pos: {r50["num_pos"]}\n'
                  f'Recall@25: Recall: {r25["ar"].item():.4f}, Num pos: {r25["num_pos"]}\nAR: {ar["ar"].item():.4f}')
        ap50 = evaluate_box_proposals_ap(proposals_list, scores_list, gt_list, iou_thresh=0.5, top_k=a.ap_top_n)
        ap25 = evaluate_box_proposals_ap(proposals_list, scores_list, gt_list, iou_thresh=0.25, top_k=a.ap_top_n)
        APs.append(ap50['ap'].item())
        print(f'AP@50: AP: {ap50["ap"].item():.4f}\nAP@25: AP: {ap25["ap"].item():.4f}')
        js['ap_50'], js['ap_25'] = ap50, ap25
        if a.mode == 'eval':
            for m in js:
                for k in js[m]:
                    if isinstance(js[m][k], torch.Tensor):
                        js[m][k] = js[m][k].tolist()
            os.makedirs(a.save_path, exist_ok=True)
            with open(os.path.join(a.save_path, 'eval.json'), 'w') as f:
                json.dump(js, f, indent=2)
        return recalls, APs

    @torch.no_grad()
    def benchmark(self):
        """Reference protocol (run_fcos.py:533-557): 10 warm-ups + 300 timed eval forwards of a randn(4,160,160,160) grid."""
        x = [torch.randn(4, 160, 160, 160, dtype=torch.float).cuda()]
        self.model.eval()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(10):
            self.model(x)
        t = np.zeros(300)
        for rep in range(300):
            start.record()
            self.model(x)
            end.record()
            torch.cuda.synchronize()
            t[rep] = start.elapsed_time(end)
        print(f'Average inference time: {t.mean():.4f} ms, std: {t.std():.4f} ms')


def main(argv=None):
    run_rpn.main(trainer_cls=Trainer, args=parse_args(argv))


if __name__ == '__main__':
    main()
