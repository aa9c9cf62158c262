"""Train / evaluate / benchmark the anchor-based NeRF-RPN on MI355X.

Command-line drop-in for the reference's ``nerf_rpn/run_rpn.py`` (flags, defaults and choices of run_rpn.py:38-143; side-effect
files ``<save_path>/{model_best.pt, epoch_N.pt, eval.json, proposals/*.npz, voxel_scores/*.npz, log/worker_k.log}``; checkpoint
dict keys ``epoch, backbone_state_dict, rpn_head_state_dict, train_args``).  What differs is underneath: the model runs on
the HIP kernels, multi-GPU training is one process per GPU with ``engine.FlatTrainer`` (flat arenas, bucketed RCCL
all-reduce overlapped with backward, fused clip + AdamW) instead of DDP + torch.optim, and the per-iteration
barrier + 4 scalar all-reduces of the reference become one fused all-reduce at logging time.

Extra flags (not in the reference): ``--dtype {fp32,bf16,bf16x3}`` (default bf16 for train/benchmark, fp32 for eval), ``--fix_obb_clip``.
"""
import argparse
import glob
import json
import logging
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from .datasets import BaseDataset, Front3DRPNDataset, GeneralRPNDataset, HypersimRPNDataset, RawScene, ScanNetRPNDataset
from .engine import FlatTrainer
from .eval import evaluate_box_proposals_ap, evaluate_box_proposals_recall
from .model.anchor import AnchorGenerator3D, RPNHead
from .model.feature_extractor import VGG_FPN, Bottleneck, ResNet_FPN_256, SwinTransformer_FPN
from .model.nerf_rpn import NeRFRegionProposalNetwork
from .model.utils import box_iou_3d

anchor_sizes = ((8,), (16,), (32,), (64,),)
aspect_ratios = (((1., 1., 1.), (1., 1., 2.), (1., 2., 2.), (1., 1., 3.), (1., 3., 3.)),) * len(anchor_sizes)
normalize_aspect_ratios = False


def build_parser():
    p = argparse.ArgumentParser(description='Train and eval the NeRF RPN baseline.')
    p.add_argument('--mode', default='train', choices=['train', 'eval', 'benchmark'])
    p.add_argument('--dataset_name', '-dn', default='hypersim', choices=['hypersim', 'front3d', 'general', 'scannet'])
    p.add_argument('--features_path', default='', help='Path to the features.')
    p.add_argument('--boxes_path', default='', help='Path to the gt boxes.')
    p.add_argument('--save_path', default='', help='Path to save the model.')
    p.add_argument('--dataset_split', default='', help='Path to the dataset split file.')
    p.add_argument('--preload', action='store_true', help='Preload the features and boxes.')
    p.add_argument('--checkpoint', default='', help='Path to the checkpoint to load.')
    p.add_argument('--load_backbone_only', action='store_true', help='Only load the backbone.')
    p.add_argument('--backbone_type', type=str, default='resnet',
                   choices=['resnet', 'vgg_AF', 'vgg_EF', 'swin_t', 'swin_s', 'swin_b', 'swin_l'])
    p.add_argument('--freeze_backbone', action='store_true', help='Freeze the backbone.')
    p.add_argument('--train_csv', default='', help='Path to the train csv. Only used if dataset_name is general.')
    p.add_argument('--val_csv', default='', help='Path to the val csv. Only used if dataset_name is general.')
    p.add_argument('--test_csv', default='', help='Path to the test csv. Only used if dataset_name is general.')
    p.add_argument('--resolution', type=int, default=160, help='The max resolution of the input features.')
    p.add_argument('--rotated_bbox', action='store_true',
                   help='If true, bbox: (N, 7), [x, y, z, w, h, d, theta]. If false, bbox: (N, 6), [xmin, ymin, zmin, xmax, ymax, zmax]')
    p.add_argument('--normalize_density', action='store_true', help='Whether to normalize the density.')
    p.add_argument('--output_proposals', action='store_true', help='Whether to output proposals during evaluation.')
    p.add_argument('--output_voxel_scores', action='store_true',
                   help='Whether to output per-voxel objectness scores during evaluation (save_path/voxel_scores).')
    p.add_argument('--filter', choices=['none', 'tp', 'fp'], default='none', help='Filter the proposal output for visualization and debugging.')
    p.add_argument('--filter_threshold', type=float, default=0.7, help='The IoU threshold for the proposal filter.')
    p.add_argument('--top_k', type=int, default=None, help='The number of proposals that will be used to calculate AP')
    p.add_argument('--rotate_prob', default=0.5, type=float, help='The probability of rotating the scene.')
    p.add_argument('--flip_prob', default=0.5, type=float, help='The probability of flipping the scene.')
    p.add_argument('--rot_scale_prob', default=0.5, type=float, help='The probability of extra rotation and scaling.')
    p.add_argument('--batch_size', default=1, type=int, help='The batch size.')
    p.add_argument('--num_epochs', default=100, type=int, help='The number of epochs to train.')
    p.add_argument('--lr', default=1e-4, type=float, help='The learning rate.')
    p.add_argument('--reg_loss_weight', default=5.0, type=float, help='The weight for balancing the regression loss.')
    p.add_argument('--reg_loss_weight_2d', default=0.0, type=float, help='The weight for balancing the 2d regression loss.')
    p.add_argument('--weight_decay', default=0.01, type=float, help='The weight decay coefficient of AdamW.')
    p.add_argument('--clip_grad_norm', default=0.1, type=float, help='The gradient clipping norm.')
    p.add_argument('--log_to_file', action='store_true', help='Whether to log to a file.')
    p.add_argument('--log_interval', default=20, type=int, help='The number of iterations to print the loss.')
    p.add_argument('--eval_interval', default=1, type=int, help='The number of epochs to evaluate.')
    p.add_argument('--keep_checkpoints', default=1, type=int, help='The number of latest checkpoints to keep.')
    p.add_argument('--wandb', action='store_true', help='Whether to use wandb for logging.')
    p.add_argument('--gpus', default='', help='The gpus to use for distributed training. If empty, uses the first available gpu. '
                                             'Data parallelism is only enabled if this is greater than one.')
    p.add_argument('--rpn_head_conv_depth', default=4, type=int, help='The number of common convolutional layers in the RPN head.')
    p.add_argument('--rpn_pre_nms_top_n_train', default=2500, type=int, help='The number of top proposals to keep before applying NMS.')
    p.add_argument('--rpn_pre_nms_top_n_test', default=2500, type=int, help='The number of top proposals to keep before applying NMS.')
    p.add_argument('--rpn_post_nms_top_n_train', default=2500, type=int, help='The number of top proposals to keep after applying NMS.')
    p.add_argument('--rpn_post_nms_top_n_test', default=2500, type=int, help='The number of top proposals to keep after applying NMS.')
    p.add_argument('--rpn_nms_thresh', default=0.3, type=float, help='The NMS threshold.')
    p.add_argument('--rpn_fg_iou_thresh', default=0.35, type=float, help='The foreground IoU threshold.')
    p.add_argument('--rpn_bg_iou_thresh', default=0.2, type=float, help='The background IoU threshold.')
    p.add_argument('--rpn_batch_size_per_mesh', default=256, type=int, help='The batch size per mesh.')
    p.add_argument('--rpn_positive_fraction', default=0.5, type=float, help='The fraction of positive proposals to use.')
    p.add_argument('--rpn_score_thresh', default=0.0, type=float, help='The score threshold.')
    p.add_argument('--reg_loss_type', choices=['smooth_l1', 'iou', 'linear_iou', 'giou', 'diou'], default='smooth_l1',
                   help='The type of regression loss to use for the RPN.')
    p.add_argument('--check_arch', action='store_true', help='Check the model architecture, then exit.')
    p.add_argument('--save_results', action='store_true', help='Save the feature maps extracted by backbone and rois for objectness')
    p.add_argument('--save_results_path', default='', help='The path to save features')
    p.add_argument('--output_all', action='store_true', help='Output proposals for train/val/test set in inference.')
    # --- extras of this implementation
    p.add_argument('--dtype', choices=['fp32', 'bf16', 'bf16x3'], default=None,
                   help='Compute dtype of the HIP conv path (default: bf16 for train/benchmark, fp32 for eval).  bf16x3 = fp32 tensors, 3x3x3 convs as '
                        'three bf16 MFMA products of split operands: fp32-grade results (the reference tolerances) at 2-3x the fp32 speed.')
    p.add_argument('--fix_obb_clip', action='store_true', help='Drop scores/levels together with out-of-grid OBBs (fixes reference quirk B3).')
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)


class Trainer:
    def __init__(self, args, rank=0, world_size=1, device_id=None, logger=None):
        self.args, self.rank, self.world_size, self.device_id = args, rank, world_size, device_id
        self.logger = logger if logger is not None else logging.getLogger()
        self.dataset = {'hypersim': HypersimRPNDataset, 'front3d': Front3DRPNDataset, 'general': GeneralRPNDataset,
                        'scannet': ScanNetRPNDataset}[args.dataset_name]
        if args.wandb and rank == 0:
            import wandb
            wandb.init(project='nerf-rpn', config=dict(vars(args)))
        self.logger.info('Constructing model...')
        self.build_backbone()
        self.anchor_generator = AnchorGenerator3D(anchor_sizes, aspect_ratios, is_normalized=normalize_aspect_ratios)
        self.rpn_head = RPNHead(self.backbone.out_channels, self.anchor_generator.num_anchors_per_location()[0],
                                args.rpn_head_conv_depth, rotate=args.rotated_bbox)
        if args.checkpoint:
            assert os.path.exists(args.checkpoint), 'The checkpoint does not exist.'
            self.logger.info(f'Loading checkpoint from {args.checkpoint}.')
            ckpt = torch.load(args.checkpoint, map_location='cpu')
            self.backbone.load_state_dict(ckpt['backbone_state_dict'])
            if not args.load_backbone_only:
                self.rpn_head.load_state_dict(ckpt['rpn_head_state_dict'])
        if args.freeze_backbone:
            for prm in self.backbone.parameters():
                prm.requires_grad = False
        self.num_bbox_digits = 6 if not args.rotated_bbox else 7
        dtype = args.dtype or ('fp32' if args.mode == 'eval' else 'bf16')
        self.model = NeRFRegionProposalNetwork(
            self.backbone, self.anchor_generator, self.rpn_head,
            rpn_pre_nms_top_n_train=args.rpn_pre_nms_top_n_train, rpn_pre_nms_top_n_test=args.rpn_pre_nms_top_n_test,
            rpn_post_nms_top_n_train=args.rpn_post_nms_top_n_train, rpn_post_nms_top_n_test=args.rpn_post_nms_top_n_test,
            rpn_nms_thresh=args.rpn_nms_thresh, rpn_fg_iou_thresh=args.rpn_fg_iou_thresh, rpn_bg_iou_thresh=args.rpn_bg_iou_thresh,
            rpn_batch_size_per_mesh=args.rpn_batch_size_per_mesh, rpn_positive_fraction=args.rpn_positive_fraction,
            rpn_score_thresh=args.rpn_score_thresh, rotated_bbox=args.rotated_bbox, reg_loss_type=args.reg_loss_type,
            compute_dtype=torch.bfloat16 if dtype == 'bf16' else torch.float32)
        if dtype == 'bf16x3':
            self.model.set_compute_dtype('bf16x3')
        self.model.rpn.fix_obb_clip = args.fix_obb_clip
        self.model.rpn.loss_2d_requires_grad = args.reg_loss_weight_2d != 0
        if args.check_arch:
            self.logger.info(self.model)
            raise SystemExit(0)
        self.model.cuda()
        self.init_datasets()

    def build_backbone(self):
        t = self.args.backbone_type
        if t == 'resnet':
            self.backbone = ResNet_FPN_256(Bottleneck, [3, 4, 6, 3], input_dim=4, is_max_pool=True)
        elif t in ('vgg_AF', 'vgg_EF'):
            self.backbone = VGG_FPN(t[-2:], 4, True, self.args.resolution)
        else:       # run_rpn.py:281-292
            swin = {'swin_t': {'embed_dim': 96, 'depths': [2, 2, 6, 2], 'num_heads': [3, 6, 12, 24]},
                    'swin_s': {'embed_dim': 96, 'depths': [2, 2, 18, 2], 'num_heads': [3, 6, 12, 24]},
                    'swin_b': {'embed_dim': 128, 'depths': [2, 2, 18, 2], 'num_heads': [3, 6, 12, 24]},
                    'swin_l': {'embed_dim': 192, 'depths': [2, 2, 18, 2], 'num_heads': [6, 12, 24, 48]}}[t]
            self.backbone = SwinTransformer_FPN(patch_size=[4, 4, 4], embed_dim=swin['embed_dim'], depths=swin['depths'],
                                                num_heads=swin['num_heads'], window_size=[4, 4, 4], stochastic_depth_prob=0.1,
                                                expand_dim=True)

    def init_datasets(self):
        a = self.args
        if not a.dataset_split and a.dataset_name != 'general':
            if a.mode == 'benchmark':
                return
            raise ValueError('The dataset split must be specified if not using general dataset.')
        if a.dataset_split:
            with np.load(a.dataset_split) as split:
                self.train_scenes, self.test_scenes, self.val_scenes = split['train_scenes'], split['test_scenes'], split['val_scenes']
                if a.output_all:
                    self.test_scenes = np.concatenate([self.train_scenes, self.test_scenes, self.val_scenes])
        if a.mode == 'eval':
            self.test_set = self._make_set(getattr(self, 'test_scenes', None), a.test_csv, augment=False)
            if self.rank == 0:
                self.logger.info(f'Loaded {len(self.test_set)} test scenes')

    def _make_set(self, scenes, csv, augment):
        a = self.args
        aug = dict(flip_prob=a.flip_prob, rotate_prob=a.rotate_prob, rot_scale_prob=a.rot_scale_prob) if augment else {}
        if a.dataset_name in ('hypersim', 'front3d'):
            ds = self.dataset(scene_list=scenes, features_path=a.features_path, boxes_path=a.boxes_path,
                              normalize_density=a.normalize_density, preload=False, **aug)
            # scenes stay in their on-disk layout on the host; alpha / layout / dtype -- and for the training set the drawn rotation,
            # flips and rotate-and-scale resampling -- happen in one pass on the GPU (ops.ingest_rgbsigma / ops.ingest_augment)
            ds.device_ingest = True
            if a.preload:
                ds.load_scene_data(preload=True)
            return ds
        if a.dataset_name == 'scannet':
            return ScanNetRPNDataset(scene_list=scenes, features_path=a.features_path, boxes_path=a.boxes_path, **aug)
        return GeneralRPNDataset(csv_path=csv, normalize_density=a.normalize_density)

    def scenes_to_device(self, rgbsigma):
        """Host scene tensors [4,W,L,H] -> device; RawScene items are finished by the ingest kernel in the compute dtype."""
        dt = getattr(self.model, 'compute_dtype', torch.float32)
        return [t.to_device(dt) if isinstance(t, RawScene) else t.cuda(non_blocking=True) for t in rgbsigma]

    def save_checkpoint(self, epoch, path):
        torch.save({'epoch': epoch, 'backbone_state_dict': self.backbone.state_dict(),
                    'rpn_head_state_dict': self.rpn_head.state_dict(), 'train_args': self.args.__dict__}, path)

    def delete_old_checkpoints(self, path, keep_latest=5):
        files = sorted(glob.glob(f'{path}/epoch_*.pt'), key=os.path.getmtime)
        for f in files[:-keep_latest] if len(files) > keep_latest else []:
            logging.info(f'Deleting old checkpoint {f}.')
            os.remove(f)

    def train_loop(self):
        a = self.args
        self.train_set = self._make_set(getattr(self, 'train_scenes', None), a.train_csv, augment=True)
        self.val_set = self._make_set(getattr(self, 'val_scenes', None), a.val_csv, augment=False)
        if self.world_size == 1:
            self.train_loader = DataLoader(self.train_set, batch_size=a.batch_size, collate_fn=BaseDataset.collate_fn, shuffle=True,
                                           num_workers=4, pin_memory=True)
        else:
            self.train_sampler = DistributedSampler(self.train_set)
            self.train_loader = DataLoader(self.train_set, batch_size=a.batch_size // self.world_size, collate_fn=BaseDataset.collate_fn,
                                           sampler=self.train_sampler, num_workers=2, pin_memory=True)
        if self.rank == 0:
            self.logger.info(f'Loaded {len(self.train_set)} training scenes, {len(self.val_set)} validation scenes')
        self.trainer = FlatTrainer(self.model, lr=a.lr, weight_decay=a.weight_decay, clip_grad_norm=a.clip_grad_norm,
                                   total_steps=a.num_epochs * len(self.train_loader))
        self.best_metric = None
        os.makedirs(a.save_path, exist_ok=True)
        for epoch in range(1, a.num_epochs + 1):
            if self.world_size > 1:
                self.train_sampler.set_epoch(epoch)
            self.train_epoch(epoch)
            if self.rank != 0:
                continue
            if epoch % a.eval_interval == 0 or epoch == a.num_epochs:
                recalls, _ = self.eval(self.val_set)
                metric = recalls[-1]
                if self.best_metric is None or metric > self.best_metric:
                    self.best_metric = metric
                    self.save_checkpoint(epoch, os.path.join(a.save_path, 'model_best.pt'))
                self.save_checkpoint(epoch, os.path.join(a.save_path, f'epoch_{epoch}.pt'))
                self.delete_old_checkpoints(a.save_path, keep_latest=a.keep_checkpoints)

    def train_epoch(self, epoch):
        a = self.args
        for i, (rgbsigma, boxes, scene_name) in enumerate(self.train_loader):
            self.model.train()
            rgbsigma = self.scenes_to_device(rgbsigma)
            # the ground truth stays on the host: the model uploads it on its target-preparation stream (NeRFRegionProposalNetwork.forward)
            _, losses, _ = self.model(rgbsigma, boxes)
            lo = losses['loss_objectness']
            lr_ = losses['loss_rpn_box_reg'] * a.reg_loss_weight
            l2 = losses['loss_rpn_box_reg_2d'] * a.reg_loss_weight_2d
            loss = lo + lr_ + l2
            loss.backward()
            lr = self.trainer.step()
            if i % a.log_interval == 0:
                vals = self.trainer.reduce_scalars(loss, lo, lr_, l2).tolist()      # one fused all-reduce, only when logging
                if self.rank == 0:
                    self.logger.info(f'Epoch {epoch} [{i}/{len(self.train_loader)}] {scene_name}  Loss: {vals[0]:.4f}  '
                                     f'Obj loss: {vals[1]:.4f}  Reg loss: {vals[2]:.4f} Reg loss 2d: {vals[3]:.4f}  lr {lr:.2e}')

    def output_proposals(self, scenes, proposals, scores, gt_boxes):
        out = os.path.join(self.args.save_path, 'proposals')
        os.makedirs(out, exist_ok=True)
        for scene, proposal, score, gt in zip(scenes, proposals, scores, gt_boxes):
            if self.args.filter != 'none':
                if proposal.shape[0] == 0 or gt is None or gt.shape[0] == 0:
                    continue
                keep = box_iou_3d(gt.cuda(), proposal.cuda()).max(dim=0)[0].cpu() > self.args.filter_threshold
                if self.args.filter == 'fp':
                    keep = ~keep
                proposal, score = proposal[keep], score[keep]
            np.savez(os.path.join(out, f'{scene}.npz'), proposal=proposal, score=score)

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
            (features, proposals, level_indexes), _, scores = self.model(rgbsigma, objectness_output_paths=paths)
            if a.save_results:
                fp, rp = os.path.join(a.save_results_path, 'features'), os.path.join(a.save_results_path, 'proposals')
                os.makedirs(fp, exist_ok=True)
                os.makedirs(rp, exist_ok=True)
                for i, s in enumerate(scenes):
                    feats = [f[i].float().cpu().numpy() for f in features]
                    np.savez(f'{fp}/{s}.npz', level_features=np.array([f.reshape(-1).astype(object) for f in feats], dtype=object),
                             resolution=[f.shape for f in feats])
                    np.savez(f'{rp}/{s}.npz', proposals=proposals[i].cpu().numpy(), level_indices=level_indexes[i].cpu().numpy())
            proposals_list += [p[:, :self.num_bbox_digits].cpu() for p in proposals]
            scores_list += [s.cpu() for s in scores]
            gt_list += [b.cpu() if b is not None else None for b in gt_boxes]
            scenes_list += list(scenes)
        if a.output_proposals:
            self.output_proposals(scenes_list, proposals_list, scores_list, gt_list)
        if gt_list[0] is None:
            return None, None
        recalls, APs, js = [], [], {}
        for limit in [300, 1000, a.rpn_post_nms_top_n_test]:
            if limit > a.rpn_post_nms_top_n_test:
                continue
            r50 = evaluate_box_proposals_recall(proposals_list, scores_list, gt_list, thresholds=torch.tensor([0.5]), limit=limit)
            r25 = evaluate_box_proposals_recall(proposals_list, scores_list, gt_list, thresholds=torch.tensor([0.25]), limit=limit)
            ar = evaluate_box_proposals_recall(proposals_list, scores_list, gt_list, thresholds=torch.arange(0.25, 1.0, 0.05), limit=limit)
            recalls.append(r50['ar'].item())
            js[f'recall_50_top_{limit}'], js[f'recall_25_top_{limit}'], js[f'recall_ar_top_{limit}'] = r50, r25, ar
            print(f'\nTop {limit} proposals:\nRecall@50: Recall: {r50["ar"].item():.4f}, Num pos: {r50["num_pos"]}\n'
                  f'Recall@25: Recall: {r25["ar"].item():.4f}, Num pos: {r25["num_pos"]}\nAR: {ar["ar"].item():.4f}')
        ap50 = evaluate_box_proposals_ap(proposals_list, scores_list, gt_list, iou_thresh=0.5, top_k=a.top_k)
        ap25 = evaluate_box_proposals_ap(proposals_list, scores_list, gt_list, iou_thresh=0.25, top_k=a.top_k)
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
        """Reference protocol (run_rpn.py:594-617): 10 warm-ups + 300 timed eval forwards of a randn(4,200,200,130) grid."""
        x = [torch.randn(4, 200, 200, 130, dtype=torch.float).cuda()]
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


def _make_logger(name, args, to_console=True):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    fmt = logging.Formatter('[%(asctime)s %(levelname)s] %(message)s')
    if to_console:
        logger.propagate = False
        h = logging.StreamHandler()
        h.setFormatter(fmt)
        h.setLevel(logging.INFO)
        logger.addHandler(h)
    if args.log_to_file:
        d = os.path.join(args.save_path, 'log')
        os.makedirs(d, exist_ok=True)
        fh = logging.FileHandler(os.path.join(d, f'{name}.log'))
        fh.setFormatter(fmt)
        fh.setLevel(logging.DEBUG)
        logger.addHandler(fh)
    return logger


def main_worker(proc, nprocs, args, gpu_ids, init_method, trainer_cls=None):
    torch.cuda.set_device(gpu_ids[proc])
    from .affinity import pin_rank
    pin = pin_rank(proc, nprocs, gpu_ids)       # each rank's enqueue thread on its own cores, next to its GPU's NUMA node (NRPN_PIN=0: off)
    if pin.get("pinned"):
        logging.info(f'rank {proc}: GPU {gpu_ids[proc]} pinned to cores {pin["cores"][0]}-{pin["cores"][1]} (NUMA node {pin["numa_node"]})')
    dist.init_process_group(backend='nccl', init_method=init_method, world_size=nprocs, rank=proc,
                            device_id=torch.device('cuda', gpu_ids[proc]))
    trainer = (trainer_cls or Trainer)(args, proc, nprocs, gpu_ids[proc], _make_logger(f'worker_{proc}', args))
    dist.barrier()
    if args.mode == 'train':
        trainer.train_loop()
    dist.destroy_process_group()


def parse_gpu_ids(spec):
    ids = []
    for token in spec.split(',') if spec else []:
        if '-' in token:
            a, b = token.split('-')
            ids.extend(range(int(a), int(b) + 1))
        else:
            ids.append(int(token))
    return ids


def main(argv=None, trainer_cls=None, args=None):
    trainer_cls = trainer_cls or Trainer
    args = args if args is not None else parse_args(argv)
    logging.basicConfig(level=logging.INFO, format='[%(asctime)s %(levelname)s] %(message)s')
    gpu_ids = parse_gpu_ids(args.gpus)
    if len(gpu_ids) <= 1:
        if len(gpu_ids) == 1:
            torch.cuda.set_device(gpu_ids[0])
        trainer = trainer_cls(args, logger=_make_logger('worker_0', args, to_console=False) if args.log_to_file else None)
        {'train': trainer.train_loop, 'eval': lambda: trainer.eval(trainer.test_set), 'benchmark': trainer.benchmark}[args.mode]()
    else:
        init_method = f'tcp://127.0.0.1:{np.random.randint(20000, 40000)}'
        logging.info(f'Using {len(gpu_ids)} processes for data parallelism, GPUs: {gpu_ids}')
        mp.spawn(main_worker, nprocs=len(gpu_ids), args=(len(gpu_ids), args, gpu_ids, init_method, trainer_cls), join=True)


if __name__ == '__main__':
    main()
