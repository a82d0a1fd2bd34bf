"""Oracle task module: ``OCTSegmentationModel`` (reference
``src/models/smp/model.py:18-200``) restated without Lightning/W&B/cv2.
TEST INFRASTRUCTURE ONLY.
"""
import numpy as np
import torch

from .losses import DiceLoss, MULTILABEL_MODE
from .metrics import get_metrics
from .nets import create_model, get_preprocessing_params


class OracleOCTSegmentationModel(torch.nn.Module):
    def __init__(self, arch, encoder_name, model_name='oracle', in_channels=3, classes=('Lumen',),
                 lr=1e-4, weight_decay=1e-4, optimizer_name='Adam', input_size=512, **kwargs):
        super().__init__()
        kwargs.pop('encoder_weights', None)
        self.model = create_model(arch, encoder_name, in_channels=in_channels, classes=len(classes))
        self.classes = list(classes)
        params = get_preprocessing_params(encoder_name)
        self.register_buffer('std', torch.tensor(params['std']).view(1, 3, 1, 1))
        self.register_buffer('mean', torch.tensor(params['mean']).view(1, 3, 1, 1))
        self.loss_fn = DiceLoss(MULTILABEL_MODE, from_logits=True)
        self.lr, self.weight_decay, self.optimizer = lr, weight_decay, optimizer_name
        self.model_name, self.input_size = model_name, input_size

    def forward(self, image):  # model.py:65-71
        return self.model((image - self.mean) / self.std)

    def training_step(self, batch, batch_idx=0):  # model.py:73-95
        img, mask = batch
        logits = self.forward(img)
        loss = self.loss_fn(logits, mask)
        pred = (logits.sigmoid() > 0.5).float()
        return {'loss': loss, 'logits': logits, 'metrics': get_metrics(mask, pred, loss)}

    def configure_optimizers(self):  # model.py:150-181
        kw = dict(lr=self.lr, weight_decay=self.weight_decay)
        if self.optimizer == 'SGD':
            return torch.optim.SGD(self.parameters(), **kw)
        if self.optimizer == 'RMSprop':
            return torch.optim.RMSprop(self.parameters(), **kw)
        if self.optimizer == 'RAdam':
            return torch.optim.RAdam(self.parameters(), **kw)
        if self.optimizer == 'Adam':
            return torch.optim.Adam(self.parameters(), **kw)
        raise ValueError(f'Unknown optimizer: {self.optimizer}')

    def predict(self, images, device='cpu'):  # model.py:183-200 (no normalisation!)
        x = torch.Tensor(images.transpose((0, 3, 1, 2)))
        y = self.model(x).cpu().detach()
        masks = (y.sigmoid() > 0.5).float()
        return masks.permute(0, 2, 3, 1).numpy().round()


_ = np
