"""UNet baseline exported by the reference next to Uformer (M1:22-140, get_arch 'UNet').  It is a
plain CNN with no custom operator on it - kept for surface parity (`from My_model_1 import UNet`)."""
import torch
import torch.nn as nn


class ConvBlock(nn.Module):
    def __init__(self, in_channel, out_channel, strides=1):
        super().__init__()
        self.strides, self.in_channel, self.out_channel = strides, in_channel, out_channel
        self.block = nn.Sequential(
            nn.Conv2d(in_channel, out_channel, kernel_size=3, stride=strides, padding=1), nn.LeakyReLU(inplace=True),
            nn.Conv2d(out_channel, out_channel, kernel_size=3, stride=strides, padding=1), nn.LeakyReLU(inplace=True))
        self.conv11 = nn.Conv2d(in_channel, out_channel, kernel_size=1, stride=strides, padding=0)

    def forward(self, x):
        return self.block(x) + self.conv11(x)


class UNet(nn.Module):
    def __init__(self, block=ConvBlock, dim=32):
        super().__init__()
        self.dim = dim
        chans = [dim, dim * 2, dim * 4, dim * 8, dim * 16]
        self.ConvBlock1 = ConvBlock(3, dim, strides=1)
        self.pool1 = nn.Conv2d(dim, dim, kernel_size=4, stride=2, padding=1)
        for i in range(1, 4):
            setattr(self, f"ConvBlock{i + 1}", block(chans[i - 1], chans[i], strides=1))
            setattr(self, f"pool{i + 1}", nn.Conv2d(chans[i], chans[i], kernel_size=4, stride=2, padding=1))
        self.ConvBlock5 = block(chans[3], chans[4], strides=1)
        for j, i in enumerate(range(6, 10)):
            cin = chans[4 - j]
            setattr(self, f"upv{i}", nn.ConvTranspose2d(cin, cin // 2, 2, stride=2))
            setattr(self, f"ConvBlock{i}", block(cin, cin // 2, strides=1))
        self.conv10 = nn.Conv2d(dim, 3, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        skips, y = [], x
        for i in range(1, 5):
            y = getattr(self, f"ConvBlock{i}")(y)
            skips.append(y)
            y = getattr(self, f"pool{i}")(y)
        y = self.ConvBlock5(y)
        for j, i in enumerate(range(6, 10)):
            y = torch.cat([getattr(self, f"upv{i}")(y), skips[3 - j]], 1)
            y = getattr(self, f"ConvBlock{i}")(y)
        return x + self.conv10(y)
