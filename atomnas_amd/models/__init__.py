"""Model builders with the reference's module paths: `models.mobilenet_supernet.Model`, `models.searched_network.Model`."""
